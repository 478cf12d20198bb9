#include "fastdeflate.h"

#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include <zlib.h>

#include <algorithm>
#include <cstring>
#include <functional>
#include <memory>

namespace reve {
namespace {

constexpr int kHashBits = 15;
constexpr int kMaxMatches = 32768;             // matches per block
constexpr size_t kProbeSpan = (size_t)1 << 16;      // a parsed block that follows blocks coded as literals only
constexpr size_t kMaxBlockSpan = (size_t)1 << 19;   // input bytes per block (statistics of image rows drift: 512 KB = ~45 rows of a 4K frame)
constexpr int kMinMatch = 4, kMaxMatch = 258;
constexpr uint32_t kWindow = 32768;

inline uint32_t load32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
inline uint64_t load64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

// RFC 1951 §3.2.5: length symbols 257..285 and distance symbols 0..29 with their extra bits
constexpr uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
constexpr uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Tables {
    uint8_t len_sym[256];     // length - 3 -> symbol - 257
    uint8_t dist_sym[512];    // d = distance - 1: d < 256 ? [d] : [256 + (d >> 7)]
    Tables()
    {
        for (int c = 0; c < 29; ++c)
            for (int l = kLenBase[c]; l < kLenBase[c] + (1 << kLenExtra[c]) && l <= kMaxMatch; ++l) len_sym[l - 3] = (uint8_t)c;
        len_sym[kMaxMatch - 3] = 28;
        for (int c = 0; c < 30; ++c)
            for (uint32_t d = kDistBase[c]; d < (uint32_t)kDistBase[c] + (1u << kDistExtra[c]); ++d) {
                const uint32_t d0 = d - 1;
                if (d0 < 256) dist_sym[d0] = (uint8_t)c;
                else dist_sym[256 + (d0 >> 7)] = (uint8_t)c;
            }
    }
};
const Tables kT;
inline int dist_symbol(uint32_t d0) { return d0 < 256 ? kT.dist_sym[d0] : kT.dist_sym[256 + (d0 >> 7)]; }

// Code lengths of an optimal prefix code limited to `maxlen` bits, then canonical codes, bit-reversed for the LSB-first
// bit stream.  Symbols with frequency 0 get length 0; at least two symbols get a code (some inflaters reject trees with one).
void build_code(const uint32_t* freq, int n, int maxlen, uint8_t* len, uint16_t* code)
{
    struct Sym { uint32_t f; int s; };
    Sym used[288];
    int m = 0;
    for (int s = 0; s < n; ++s)
        if (freq[s]) used[m++] = {freq[s], s};
    for (int s = 0; m < 2 && s < n; ++s) {       // fewer than two symbols in use: give unused ones a code
        bool have = false;
        for (int i = 0; i < m; ++i) have = have || used[i].s == s;
        if (!have) used[m++] = {1, s};
    }
    std::sort(used, used + m, [](const Sym& a, const Sym& b) { return a.f != b.f ? a.f < b.f : a.s < b.s; });
    // Huffman tree by the two-queue method: leaves 0..m-1 in ascending weight, internal nodes m..2m-2 in creation order
    uint64_t w[2 * 288];
    int parent[2 * 288], depth[2 * 288];
    for (int i = 0; i < m; ++i) w[i] = used[i].f;
    int leaf = 0, inode = m, next = m;
    auto pop = [&]() { return (leaf < m && (inode >= next || w[leaf] <= w[inode])) ? leaf++ : inode++; };
    while (next < 2 * m - 1) {
        const int a = pop(), b = pop();
        w[next] = w[a] + w[b];
        parent[a] = parent[b] = next;
        ++next;
    }
    depth[2 * m - 2] = 0;
    for (int i = 2 * m - 3; i >= 0; --i) depth[i] = depth[parent[i]] + 1;
    // lengths above the limit: move them to the limit and repair the Kraft sum by lengthening the longest shorter codes
    int num[300] = {};
    for (int i = 0; i < m; ++i) num[depth[i] < 299 ? depth[i] : 299]++;
    bool over = false;
    for (int l = maxlen + 1; l < 300; ++l)
        if (num[l]) { num[maxlen] += num[l]; num[l] = 0; over = true; }
    if (over) {
        uint64_t total = 0;
        for (int l = 1; l <= maxlen; ++l) total += (uint64_t)num[l] << (maxlen - l);
        while (total != ((uint64_t)1 << maxlen)) {
            num[maxlen]--;
            for (int l = maxlen - 1; l > 0; --l)
                if (num[l]) { num[l]--; num[l + 1] += 2; break; }
            total--;
        }
    }
    std::memset(len, 0, (size_t)n);
    for (int l = maxlen, i = 0; l >= 1; --l)     // ascending weight = descending length
        for (int k = 0; k < num[l]; ++k) len[used[i++].s] = (uint8_t)l;
    uint16_t next_code[16] = {};
    {
        int cnt[16] = {};
        for (int s = 0; s < n; ++s) cnt[len[s]]++;
        cnt[0] = 0;
        uint16_t c = 0;
        for (int l = 1; l <= maxlen; ++l) { c = (uint16_t)((c + cnt[l - 1]) << 1); next_code[l] = c; }
    }
    for (int s = 0; s < n; ++s) {
        uint16_t c = 0;
        if (len[s]) {
            uint16_t v = next_code[len[s]]++;
            for (int b = 0; b < len[s]; ++b) { c = (uint16_t)((c << 1) | (v & 1)); v >>= 1; }
        }
        code[s] = c;
    }
}

struct BitWriter {
    uint8_t* p;
    uint64_t acc = 0;
    int n = 0;
    inline void put(uint32_t v, int c)       // c <= 32; at most 31 bits pending before the call
    {
        acc |= (uint64_t)v << n;
        n += c;
        if (n >= 32) {
            const uint32_t lo = (uint32_t)acc;
            std::memcpy(p, &lo, 4);
            p += 4;
            acc >>= 32;
            n -= 32;
        }
    }
    // up to 56 bits at once (the literal emitter: several codes joined); at most 7 bits pending before the call
    inline void put_wide(uint64_t v, int c)
    {
        acc |= v << n;
        n += c;
        std::memcpy(p, &acc, 8);
        p += n >> 3;
        acc >>= n & ~7;
        n &= 7;
    }
    void align()                              // to a byte boundary, zero-padded
    {
        while (n > 0) { *p++ = (uint8_t)acc; acc >>= 8; n -= 8; }
        acc = 0;
        n = 0;
    }
};

struct Block {
    // the parse of src[start, end): match k is preceded by lit_run[k] literal bytes (taken from src when the block is written);
    // what follows the last match up to `end` is literals too
    uint32_t lit_run[kMaxMatches];
    uint32_t match[kMaxMatches];              // (length - 3) << 15 | (distance - 1)
    int nmatch = 0;
    uint32_t lfreq[288], dfreq[32];
    void reset()
    {
        nmatch = 0;
        std::memset(lfreq, 0, sizeof(lfreq));
        std::memset(dfreq, 0, sizeof(dfreq));
    }
};

// Joined codes of byte PAIRS for blocks without matches (grain, noise that still codes below 8 bits): one lookup and one shift
// per two literals.  256 KB per encoder thread, rebuilt per block (65,536 entries: ~0.1 ms against the ~0.5 ms a block takes).
struct PairTable { uint32_t e[65536]; };      // code (<= 24 bits) | length << 24

// The literal-only emitter: four literals per 8-byte write (4 x 12 bits + 7 pending <= 56), two table lookups.  One such stream
// is ONE dependency chain through the writer's bit count (shift, or, store, shift, and: ~8 cycles per four bytes), so a block is
// coded as SEVERAL streams at once — its first part into the output, the others into a scratch buffer, the chains
// interleaved in one loop — and those are then appended behind the first with a word-wise shift (append_bits: no chain,
// ~0.1 ns per byte).  With BMI2's three-operand variable shifts (shlx / shrx) the chain is a third shorter, so that build is chosen
// at run time.  A 4K frame of grain: 25 ms (one stream) -> 17 (BMI2) -> 11 (two streams).
struct LitWriter {             // BitWriter's state by value, for two of them to live in registers
    uint8_t* p;
    uint64_t acc;
    int n;
};
// (the writers' state lives in LOCAL variables inside the loop: a store through a byte pointer may alias anything that is reachable
// through a reference, and would force every field to be re-read after every write)
#define REVE_PUT_WIDE(P, ACC, N, v, c)    \
    do {                                  \
        ACC |= (uint64_t)(v) << N;        \
        N += (c);                         \
        std::memcpy(P, &ACC, 8);          \
        P += N >> 3;                      \
        ACC >>= N & ~7;                   \
        N &= 7;                           \
    } while (0)
#ifndef REVE_EMIT_STREAMS
#define REVE_EMIT_STREAMS 2      // (3 and 4 measured 6 % and 10 % SLOWER per frame than 2: registers, and the appended share grows)
#endif
constexpr int kEmitStreams = REVE_EMIT_STREAMS;
// stream s codes blk[s * part, (s + 1) * part); the last one also the rest up to span
#define REVE_EMITN_BODY                                                                                                   \
    uint8_t* p[NS];                                                                                                       \
    uint64_t acc[NS];                                                                                                     \
    int n[NS];                                                                                                            \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) { p[s] = W[s].p; acc[s] = W[s].acc; n[s] = W[s].n; }                   \
    const uint8_t* q = blk;                                                                                               \
    size_t k = part;                                                                                                      \
    for (; k >= 4; k -= 4, q += 4) {                                                                                      \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) {                                                                  \
            uint16_t a, c;                                                                                                \
            std::memcpy(&a, q + s * part, 2);                                                                             \
            std::memcpy(&c, q + s * part + 2, 2);                                                                         \
            const uint32_t x = e[a], y = e[c];                                                                            \
            const int l = (int)(x >> 24);                                                                                 \
            const uint64_t v = (uint64_t)(x & 0xffffff) | ((uint64_t)(y & 0xffffff) << l);                                \
            REVE_PUT_WIDE(p[s], acc[s], n[s], v, l + (int)(y >> 24));                                                     \
        }                                                                                                                 \
    }                                                                                                                     \
    for (; k; --k, ++q)                                                                                                   \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) { const uint32_t t = lit[q[s * part]]; REVE_PUT_WIDE(p[s], acc[s], n[s], t & 0xffff, (int)(t >> 16)); } \
    for (const uint8_t* r = blk + NS * part; r < blk + span; ++r) { const uint32_t t = lit[*r]; REVE_PUT_WIDE(p[NS - 1], acc[NS - 1], n[NS - 1], t & 0xffff, (int)(t >> 16)); } \
    _Pragma("unroll") for (int s = 0; s < NS; ++s) { W[s].p = p[s]; W[s].acc = acc[s]; W[s].n = n[s]; }
template <int NS>
void emit_streams(LitWriter (&W)[NS], const uint32_t* e, const uint32_t* lit, const uint8_t* blk, size_t part, size_t span) { REVE_EMITN_BODY }
#if defined(__x86_64__)
template <int NS>
__attribute__((target("bmi2"))) void emit_streams_bmi2(LitWriter (&W)[NS], const uint32_t* e, const uint32_t* lit, const uint8_t* blk, size_t part, size_t span) { REVE_EMITN_BODY }
#endif
#undef REVE_EMITN_BODY

// appends the first `nbits` bits of src (LSB first; src is readable up to the next multiple of 8 bytes + 8) to the stream: words
// shifted by the writer's pending bit count
void append_bits(BitWriter& bw, const uint8_t* src, size_t nbits)
{
    while (bw.n >= 8) { *bw.p++ = (uint8_t)bw.acc; bw.acc >>= 8; bw.n -= 8; }
    const int sh = bw.n;
    uint64_t carry = bw.acc;                         // `sh` pending bits
    const size_t words = nbits / 64;
    uint8_t* o = bw.p;
    for (size_t k = 0; k < words; ++k, o += 8) {
        const uint64_t v = load64(src + 8 * k);
        const uint64_t w = carry | (v << sh);
        std::memcpy(o, &w, 8);
        carry = sh ? v >> (64 - sh) : 0;
    }
    bw.p = o;
    bw.acc = carry;
    bw.n = sh;
    size_t rest = nbits - words * 64;                // < 64 bits left, at src + 8 * words
    const uint8_t* q = src + 8 * words;
    while (rest) {
        const int take = rest < 24 ? (int)rest : 24;
        uint32_t v = load32(q) & (take == 32 ? 0xffffffffu : ((1u << take) - 1));
        bw.put(v, take);
        q += 3;
        rest -= (size_t)take;
        if (take < 24) break;
    }
}

// one deflate block for src[start, end): dynamic Huffman codes, or stored where that is not larger.  `lit_only`: the block holds
// no match (its literal code is then limited to 12 bits so that four literals fit one 56-bit write).  Returns what the block
// would have cost, relative to what it did cost, had its matches been coded as literals (1.0 for blocks without matches): the
// caller's measure of what the match search buys.
double write_block(BitWriter& bw, Block& b, const uint8_t* blk, size_t span, bool final, bool lit_only = false)
{
    b.lfreq[256] = 1;
    uint8_t llen[288], dlen[32], cllen[19];
    uint16_t lcode[288], dcode[32], clcode[19];
    build_code(b.lfreq, 286, lit_only ? 12 : 15, llen, lcode);
    build_code(b.dfreq, 30, 15, dlen, dcode);
    int hlit = 286, hdist = 30;
    while (hlit > 257 && !llen[hlit - 1]) --hlit;
    while (hdist > 1 && !dlen[hdist - 1]) --hdist;
    // the two length sequences as one, run-length coded with the symbols 16 (repeat previous 3-6), 17 (zeros 3-10), 18 (zeros 11-138)
    uint8_t seq[288 + 32];
    std::memcpy(seq, llen, (size_t)hlit);
    std::memcpy(seq + hlit, dlen, (size_t)hdist);
    const int nseq = hlit + hdist;
    uint8_t rsym[320], rext[320];
    int nr = 0;
    uint32_t clfreq[19] = {};
    for (int i = 0; i < nseq;) {
        int run = 1;
        while (i + run < nseq && seq[i + run] == seq[i]) ++run;
        const int v = seq[i];
        i += run;
        if (v == 0) {
            while (run >= 11) { const int r = run < 138 ? run : 138; rsym[nr] = 18; rext[nr++] = (uint8_t)(r - 11); run -= r; }
            if (run >= 3) { rsym[nr] = 17; rext[nr++] = (uint8_t)(run - 3); run = 0; }
        } else {
            rsym[nr] = (uint8_t)v; rext[nr++] = 0; --run;
            while (run >= 3) { const int r = run < 6 ? run : 6; rsym[nr] = 16; rext[nr++] = (uint8_t)(r - 3); run -= r; }
        }
        while (run-- > 0) { rsym[nr] = (uint8_t)v; rext[nr++] = 0; }
    }
    for (int i = 0; i < nr; ++i) clfreq[rsym[i]]++;
    build_code(clfreq, 19, 7, cllen, clcode);
    int hclen = 19;
    while (hclen > 4 && !cllen[kClOrder[hclen - 1]]) --hclen;
    uint64_t bits = 3 + 5 + 5 + 4 + 3 * (uint64_t)hclen;
    for (int i = 0; i < nr; ++i) bits += cllen[rsym[i]] + (rsym[i] == 16 ? 2 : rsym[i] == 17 ? 3 : rsym[i] == 18 ? 7 : 0);
    for (int s = 0; s < 286; ++s) bits += (uint64_t)b.lfreq[s] * (llen[s] + (s > 256 ? kLenExtra[s - 257] : 0));
    for (int s = 0; s < 30; ++s) bits += (uint64_t)b.dfreq[s] * (dlen[s] + kDistExtra[s]);
    const uint64_t stored_bits = 8 * ((uint64_t)span + 5 * ((span + 65534) / 65535 + (span == 0)) + 1);
    // the literals' mean code length, applied to the bytes the matches cover: the block as literals only
    double as_literals = 1.0;
    if (b.nmatch) {
        uint64_t lit_bits = 0, lit_n = 0;
        for (int v = 0; v < 256; ++v) { lit_bits += (uint64_t)b.lfreq[v] * llen[v]; lit_n += b.lfreq[v]; }
        const double mean = lit_n ? (double)lit_bits / (double)lit_n : 8.0;
        as_literals = std::min((double)stored_bits, mean * (double)span) / (double)std::min(bits, stored_bits);
    }
    auto stored = [&]() {
        size_t at = 0;
        do {
            const size_t k = std::min<size_t>(span - at, 65535);
            bw.put((final && at + k == span) ? 1 : 0, 3);     // BFINAL, BTYPE = 00
            bw.align();
            const uint32_t hdr = (uint32_t)k | ((uint32_t)(k ^ 0xffff) << 16);
            std::memcpy(bw.p, &hdr, 4);
            if (k) std::memcpy(bw.p + 4, blk + at, k);
            bw.p += 4 + k;
            at += k;
        } while (at < span);
    };
    if (bits >= stored_bits) {
        stored();
        b.reset();
        return as_literals;
    }
    const BitWriter block_start = bw;                        // (the sampled-histogram blocks may have to be taken back, below)
    bw.put((final ? 1 : 0) | (2 << 1), 3);                   // BFINAL, BTYPE = 10
    bw.put((uint32_t)(hlit - 257), 5);
    bw.put((uint32_t)(hdist - 1), 5);
    bw.put((uint32_t)(hclen - 4), 4);
    for (int i = 0; i < hclen; ++i) bw.put(cllen[kClOrder[i]], 3);
    for (int i = 0; i < nr; ++i) {
        bw.put(clcode[rsym[i]], cllen[rsym[i]]);
        if (rsym[i] >= 16) bw.put(rext[i], rsym[i] == 16 ? 2 : rsym[i] == 17 ? 3 : 7);
    }
    uint32_t lit[256];                                       // code | length << 16
    for (int v = 0; v < 256; ++v) lit[v] = lcode[v] | ((uint32_t)llen[v] << 16);
    auto literals = [&](const uint8_t* p, size_t k) {        // two per put: 2 x 15 bits at most
        for (; k >= 2; k -= 2, p += 2) {
            const uint32_t e0 = lit[p[0]], e1 = lit[p[1]];
            bw.put((e0 & 0xffff) | ((e1 & 0xffff) << (e0 >> 16)), (int)((e0 >> 16) + (e1 >> 16)));
        }
        if (k) bw.put(lit[*p] & 0xffff, (int)(lit[*p] >> 16));
    };
    if (lit_only && b.nmatch == 0 && span >= 64) {
        static thread_local std::unique_ptr<PairTable> pt;
        if (!pt) pt.reset(new PairTable);
        uint32_t* const e = pt->e;
        for (int hi = 0; hi < 256; ++hi) {            // index = p[0] | p[1] << 8 (a little-endian 16-bit load)
            const uint32_t c1 = lit[hi] & 0xffff, l1 = lit[hi] >> 16;
            for (int lo = 0; lo < 256; ++lo) {
                const uint32_t l0 = lit[lo] >> 16;
                e[(hi << 8) | lo] = ((lit[lo] & 0xffff) | (c1 << l0)) | ((l0 + l1) << 24);
            }
        }
        // (the 32-bit writer may hold up to 31 bits: drain whole bytes so that the wide writes' "at most 7 pending" holds)
        while (bw.n >= 8) { *bw.p++ = (uint8_t)bw.acc; bw.acc >>= 8; bw.n -= 8; }
        static thread_local std::vector<uint8_t> scratch;          // the later streams' bits: at most 12 per literal
        constexpr int NS = kEmitStreams;
        const size_t part = span / NS, cap1 = (span - (NS - 1) * part) * 3 / 2 + 64;
        if (scratch.size() < (NS - 1) * cap1) scratch.resize((NS - 1) * cap1);
        LitWriter W[NS];
        W[0] = {bw.p, bw.acc, bw.n};
        for (int k = 1; k < NS; ++k) W[k] = {scratch.data() + (size_t)(k - 1) * cap1, 0, 0};
#if defined(__x86_64__)
        static const bool bmi2 = __builtin_cpu_supports("bmi2");
        if (bmi2) emit_streams_bmi2<NS>(W, e, lit, blk, part, span);
        else
#endif
            emit_streams<NS>(W, e, lit, blk, part, span);
        bw.p = W[0].p; bw.acc = W[0].acc; bw.n = W[0].n;
        for (int k = 1; k < NS; ++k) {
            uint8_t* const base = scratch.data() + (size_t)(k - 1) * cap1;
            std::memcpy(W[k].p, &W[k].acc, 8);                      // (its pending bits, so that append_bits reads them from memory)
            append_bits(bw, base, (size_t)(W[k].p - base) * 8 + (size_t)W[k].n);
        }
        while (bw.n >= 8) { *bw.p++ = (uint8_t)bw.acc; bw.acc >>= 8; bw.n -= 8; }
        bw.put_wide(lcode[256], llen[256]);
        // `bits` above came from a SAMPLED histogram (compress_core), so it is an estimate: bytes the sample did not see can make the
        // block longer than its stored form, up to 12 bits per byte.  The size the caller's buffer is laid out for is the stored
        // one (plus half a block of room for exactly this attempt), so a block that came out longer is taken back and stored.
        if ((uint64_t)(bw.p - block_start.p) * 8 + (uint64_t)bw.n > stored_bits + (uint64_t)block_start.n) {
            bw = block_start;
            stored();
        }
        b.reset();
        return as_literals;
    }
    const uint8_t* q = blk;
    for (int i = 0; i < b.nmatch; ++i) {
        literals(q, b.lit_run[i]);
        q += b.lit_run[i];
        const uint32_t t = b.match[i], l3 = t >> 15, d0 = t & 0x7fff;
        const int ls = kT.len_sym[l3], ds = dist_symbol(d0);
        bw.put(lcode[257 + ls] | ((l3 + 3 - kLenBase[ls]) << llen[257 + ls]), llen[257 + ls] + kLenExtra[ls]);
        bw.put(dcode[ds] | ((d0 + 1 - kDistBase[ds]) << dlen[ds]), dlen[ds] + kDistExtra[ds]);
        q += l3 + 3;
    }
    literals(q, (size_t)(blk + span - q));
    bw.put(lcode[256], llen[256]);
    b.reset();
    return as_literals;
}

#if defined(__x86_64__)
__attribute__((target("avx2"))) uint32_t adler32_avx2(uint32_t adler, const uint8_t* p, size_t n)
{
    uint32_t s1 = adler & 0xffff, s2 = adler >> 16;
    const __m256i weights = _mm256_setr_epi8(32, 31, 30, 29, 28, 27, 26, 25, 24, 23, 22, 21, 20, 19, 18, 17, 16, 15, 14, 13, 12, 11, 10, 9, 8, 7, 6,
                                             5, 4, 3, 2, 1);
    const __m256i ones = _mm256_set1_epi16(1), zero = _mm256_setzero_si256();
    while (n >= 32) {
        const size_t blk = std::min<size_t>(n, 5536);        // a multiple of 32 below zlib's NMAX: the 32-bit sums cannot overflow
        const size_t take = blk & ~(size_t)31;
        n -= take;
        __m256i v1 = zero, v2 = zero, v1sum = zero;
        for (size_t k = 0; k < take; k += 32, p += 32) {
            const __m256i v = _mm256_loadu_si256((const __m256i*)p);
            v1sum = _mm256_add_epi32(v1sum, v1);              // s1 before this chunk, summed over the chunks
            v1 = _mm256_add_epi32(v1, _mm256_sad_epu8(v, zero));
            v2 = _mm256_add_epi32(v2, _mm256_madd_epi16(_mm256_maddubs_epi16(v, weights), ones));
        }
        alignas(32) uint32_t a[8], bsum[8], c[8];
        _mm256_store_si256((__m256i*)a, v1);
        _mm256_store_si256((__m256i*)bsum, v2);
        _mm256_store_si256((__m256i*)c, v1sum);
        uint64_t h1 = 0, h2 = 0, h3 = 0;
        for (int i = 0; i < 8; ++i) { h1 += a[i]; h2 += bsum[i]; h3 += c[i]; }
        s2 = (uint32_t)((s2 + (uint64_t)s1 * take + h2 + 32 * h3) % 65521);
        s1 = (uint32_t)((s1 + h1) % 65521);
    }
    for (; n; --n) { s1 += *p++; s2 += s1; }
    return ((s2 % 65521) << 16) | (s1 % 65521);
}
#endif

#if defined(__x86_64__)
// CRC-32 (the PNG / zlib polynomial, reflected) by carry-less multiplication: four 128-bit lanes folded 64 bytes at a time, then
// to one lane, to 64 bits and Barrett-reduced (Gopal et al., "Fast CRC Computation for Generic Polynomials Using PCLMULQDQ");
// the fold constants are x^(512+32), x^(512-32), x^(128+32), x^(128-32), x^64 mod P, bit-reflected, then P' and mu.  `state` is the
// inverted running CRC; n >= 64 and a multiple of 16.  tests: compared with zlib's crc32 on random buffers of every length.
__attribute__((target("pclmul,sse4.1"))) uint32_t crc32_clmul(uint32_t state, const uint8_t* p, size_t n)
{
    const __m128i k1k2 = _mm_set_epi64x(0x01c6e41596, 0x0154442bd4), k3k4 = _mm_set_epi64x(0x00ccaa009e, 0x01751997d0);
    const __m128i k5 = _mm_set_epi64x(0, 0x0163cd6124), poly = _mm_set_epi64x(0x01f7011641, 0x01db710641);
    __m128i x1 = _mm_loadu_si128((const __m128i*)p), x2 = _mm_loadu_si128((const __m128i*)(p + 16));
    __m128i x3 = _mm_loadu_si128((const __m128i*)(p + 32)), x4 = _mm_loadu_si128((const __m128i*)(p + 48));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)state));
    p += 64;
    n -= 64;
    for (; n >= 64; n -= 64, p += 64) {
        const __m128i a1 = _mm_clmulepi64_si128(x1, k1k2, 0x00), a2 = _mm_clmulepi64_si128(x2, k1k2, 0x00);
        const __m128i a3 = _mm_clmulepi64_si128(x3, k1k2, 0x00), a4 = _mm_clmulepi64_si128(x4, k1k2, 0x00);
        x1 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x1, k1k2, 0x11), a1), _mm_loadu_si128((const __m128i*)p));
        x2 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x2, k1k2, 0x11), a2), _mm_loadu_si128((const __m128i*)(p + 16)));
        x3 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x3, k1k2, 0x11), a3), _mm_loadu_si128((const __m128i*)(p + 32)));
        x4 = _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x4, k1k2, 0x11), a4), _mm_loadu_si128((const __m128i*)(p + 48)));
    }
#define REVE_CRC_FOLD(x, next) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k3k4, 0x11), _mm_clmulepi64_si128(x, k3k4, 0x00)), next)
    x1 = REVE_CRC_FOLD(x1, x2);
    x1 = REVE_CRC_FOLD(x1, x3);
    x1 = REVE_CRC_FOLD(x1, x4);
    for (; n >= 16; n -= 16, p += 16) x1 = REVE_CRC_FOLD(x1, _mm_loadu_si128((const __m128i*)p));
#undef REVE_CRC_FOLD
    const __m128i mask32 = _mm_setr_epi32(~0, 0, ~0, 0);
    __m128i t = _mm_clmulepi64_si128(x1, k3k4, 0x10);
    x1 = _mm_xor_si128(_mm_srli_si128(x1, 8), t);                    // 128 -> 64 bits
    t = _mm_srli_si128(x1, 4);
    x1 = _mm_xor_si128(_mm_clmulepi64_si128(_mm_and_si128(x1, mask32), k5, 0x00), t);
    t = _mm_clmulepi64_si128(_mm_and_si128(x1, mask32), poly, 0x10);   // Barrett
    t = _mm_clmulepi64_si128(_mm_and_si128(t, mask32), poly, 0x00);
    return (uint32_t)_mm_extract_epi32(_mm_xor_si128(x1, t), 1);
}
#endif

}  // namespace

uint32_t fast_crc32(uint32_t crc, const uint8_t* buf, size_t n)
{
#if defined(__x86_64__)
    static const bool clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1");
    if (clmul && n >= 64) {
        const size_t k = n & ~(size_t)15;
        crc = ~crc32_clmul(~crc, buf, k);
        buf += k;
        n -= k;
    }
#endif
    while (n) {                                  // (zlib's takes a 32-bit length)
        const size_t k = std::min<size_t>(n, (size_t)1 << 30);
        crc = (uint32_t)crc32(crc, buf, (uInt)k);
        buf += k;
        n -= k;
    }
    return crc;
}

uint32_t fast_adler32(uint32_t adler, const uint8_t* buf, size_t n)
{
#if defined(__x86_64__)
    static const bool avx2 = __builtin_cpu_supports("avx2");
    if (avx2) return adler32_avx2(adler, buf, n);
#endif
    while (n) {                                  // (zlib's takes a 32-bit length)
        const size_t k = std::min<size_t>(n, (size_t)1 << 30);
        adler = (uint32_t)adler32(adler, buf, (uInt)k);
        buf += k;
        n -= k;
    }
    return adler;
}

namespace {

// Where the encoder's bytes come from.  Contiguous: the caller's buffer.  Rows: a window the encoder owns — 32 KB of history, the
// block being parsed, some rows of lookahead — that a callback fills a few rows at a time: a PNG encoder filters its scanlines
// straight into it, so the 25 MB of a 4K frame's filtered rows never travel to memory and back (that round trip was 11 of a
// frame's 60 ms), and the bytes are in cache when the histogram and the emitter read them.
struct Contiguous {
    const uint8_t* data;
    size_t total;
    const uint8_t* at(size_t pos) const { return data + pos; }
    size_t ensure(size_t) { return total; }
    void block_done(size_t) {}
};

struct RowWindow {
    std::vector<uint8_t>& buf;
    size_t total, row_bytes;
    const std::function<void(uint8_t*, size_t)>& fill;      // the next `rows` rows (row_bytes each) to dst
    size_t base = 0, filled = 0;                                 // stream position of buf[0]; end of what has been produced
    static constexpr size_t kLook = 1024;                        // a match and the widest load past the parse position
    RowWindow(std::vector<uint8_t>& b, size_t t, size_t r, const std::function<void(uint8_t*, size_t)>& f) : buf(b), total(t), row_bytes(r), fill(f)
    {
        // history (which may lag a row behind) + a block and its last match + the lookahead, rounded up to whole rows
        const size_t cap = kWindow + kMaxBlockSpan + 4096 + kLook + 3 * row_bytes;
        if (buf.size() < cap) buf.resize(cap);
    }
    const uint8_t* at(size_t pos) const { return buf.data() + (pos - base); }
    // makes [.., min(total, pos + kLook)) available; returns the end of what is
    size_t ensure(size_t pos)
    {
        while (filled < total && filled < pos + kLook) {
            size_t rows = (buf.size() - 64 - (filled - base)) / row_bytes;
            rows = std::min(rows, std::max<size_t>(1, 65536 / row_bytes));       // a few rows at a time: they stay in L1 / L2 for the parse
            rows = std::min(rows, (total - filled) / row_bytes);
            if (rows == 0) break;                                                  // (cannot happen: the capacity leaves two rows beyond any block)
            fill(buf.data() + (filled - base), rows);
            filled += rows * row_bytes;
        }
        return filled;
    }
    // a block ended at `pos`: everything older than the match window can go
    void block_done(size_t pos)
    {
        if (pos < base + kWindow + row_bytes) return;
        const size_t nb = pos - kWindow;
        std::memmove(buf.data(), buf.data() + (nb - base), filled - nb);
        base = nb;
    }
};

template <class Src>
size_t compress_core(Src& S, std::vector<uint8_t>& out, size_t offset)
{
    const size_t n = S.total;
    // worst case: every block stored, 5 bytes per 65535 and one of padding per block of >= 32 K tokens, header, trailer — and half a
    // block beyond that: a literal-only block is first WRITTEN with its 12-bit codes (at most 1.5 bytes per byte) and taken back
    // if that came out longer than storing it (write_block)
    const size_t cap = offset + n + n / 2048 + 4096 + std::min(n, kMaxBlockSpan) / 2 + 512;
    if (out.size() < cap) out.resize(cap);
    static thread_local std::vector<uint32_t> head_v;
    static thread_local std::unique_ptr<Block> blk;
    if (!blk) blk.reset(new Block);              // 130 KB per encoder thread
    head_v.assign((size_t)1 << kHashBits, 0);
    uint32_t* const head = head_v.data();
    Block& b = *blk;
    b.reset();
    BitWriter bw;
    bw.p = out.data() + offset;
    *bw.p++ = 0x78;                              // deflate, 32 K window
    *bw.p++ = 0x01;                              // fastest algorithm, no dictionary; 0x7801 is a multiple of 31
    size_t i = 0, start = 0, lit_start = 0;    // lit_start: first byte of the literal run in progress
    uint32_t misses = 0, last_d0 = 0xffffffffu;
    const size_t safe = n >= 16 ? n - 16 : 0;    // below `safe` every 4-byte and 8-byte load stays inside the stream
    int lit_only = 0;                            // blocks still to be coded without looking for matches
    bool probing = false;                        // the block being parsed follows such a stretch: kept short
    size_t avail = S.ensure(0);                  // end of the bytes that can be read
    while (i < n) {
        if (lit_only > 0 && i == start) {
            // Low-entropy noise (upscaled grain: residuals of a few small values) is full of chance repeats of 4-6 bytes
            // that cost more bits than the literals they replace and a probe each; after a block like that the next ones
            // are coded with Huffman codes alone (a byte histogram), then one block is parsed again to see what the rows
            // look like now.
            // The byte histogram is SAMPLED — 64 bytes of every 256: a code built from a quarter of half a megabyte is within a
            // fraction of a percent of the exact one — and every byte value gets a count of at least one, so that whatever the
            // unsampled bytes hold can be coded.
            const size_t e = std::min(n, i + kMaxBlockSpan);
            avail = S.ensure(e);
            const uint8_t* const q = S.at(i);
            const size_t span = e - i;
            uint32_t h4[4][256] = {};
            size_t j = 0;
            uint32_t words = 0, repeats = 0;       // sampled 8-byte words / those equal to the bytes one RGB pixel earlier
            for (; j + 256 <= span; j += 256) {
                for (size_t t = j; t < j + 64; t += 4) { h4[0][q[t]]++; h4[1][q[t + 1]]++; h4[2][q[t + 2]]++; h4[3][q[t + 3]]++; }
                for (size_t t = j + 8; t < j + 64; t += 8, ++words) repeats += load64(q + t) == load64(q + t - 3);
            }
            for (; j < span; ++j) h4[0][q[j]]++;
            for (int v = 0; v < 256; ++v) b.lfreq[v] += 1 + 4 * (h4[0][v] + h4[1][v] + h4[2][v] + h4[3][v]);
            if (4 * repeats > words) {
                // a quarter of the sampled words repeat the pixel before them (flat rows after the Up filter: runs of zeros) — what the
                // match search codes ten times smaller and faster than literals.  The stretch ends here (a frame of grain with flat
                // bands below it).  Grain whose commonest residual is 0 half of the time still has no runs: 0.5^8 of its words repeat.
                b.reset();
                lit_only = 0;
                probing = false;
                continue;
            }
            i = e;
            if (--lit_only == 0) probing = true;
            write_block(bw, b, q, span, i == n, true);
            if (i == n) goto trailer;
            start = lit_start = i;
            S.block_done(i);
            continue;
        }
        if (i + 300 > avail && avail < n) avail = S.ensure(i);
        size_t len = 0;
        uint32_t d0 = 0;
        if (i < safe && i + 16 <= avail) {          // (`avail` never limits in practice: the window keeps a kilobyte ahead of the parse)
            const uint8_t* const c = S.at(i);
            const uint32_t v = load32(c);
            const uint32_t h = (v * 2654435761u) >> (32 - kHashBits);
            uint32_t cand = head[h];
            head[h] = (uint32_t)i;
            // the previous match's distance first: runs and periodic patterns keep ONE distance (a one-bit symbol) instead of
            // the distance back to the previous token's start, which is what the hash table holds
            const bool rep = last_d0 < i && load32(c - last_d0 - 1) == v;
            if (rep) cand = (uint32_t)i - last_d0 - 1;
            d0 = (uint32_t)i - cand - 1;
            if (d0 < kWindow && load32(c - d0 - 1) == v) {
                const size_t maxl = std::min<size_t>(kMaxMatch, avail - i);
                len = kMinMatch;
                const uint8_t* a = c - d0 - 1;
                for (;;) {
                    if (len + 8 > maxl || i + len + 8 > avail) {
                        while (len < maxl && a[len] == c[len]) ++len;
                        break;
                    }
                    const uint64_t x = load64(a + len) ^ load64(c + len);
                    if (x) { len += (size_t)__builtin_ctzll(x) >> 3; break; }
                    len += 8;
                }
                // a short match far away costs more bits than the literals it replaces (noisy rows are full of them)
                if (!rep && len < (size_t)(kMinMatch + (d0 >= 32) + (d0 >= 1024) + (d0 >= 8192))) len = 0;
            }
        }
        if (len) {
            b.lit_run[b.nmatch] = (uint32_t)(i - lit_start);
            b.match[b.nmatch++] = ((uint32_t)(len - 3) << 15) | d0;
            b.lfreq[257 + kT.len_sym[len - 3]]++;
            b.dfreq[dist_symbol(d0)]++;
            i += len;
            lit_start = i;
            misses = 0;
            last_d0 = d0;
        } else {
            // a literal; after a stretch without matches more than one per probe (incompressible data is stepped over)
            size_t k = 1 + (misses >> 5);
            if (k > 32) k = 32;
            if (k > avail - i) k = avail - i;
            ++misses;
            const uint8_t* const c = S.at(i);
            for (size_t j = 0; j < k; ++j) b.lfreq[c[j]]++;
            i += k;
        }
        if ((b.nmatch == kMaxMatches || i - start >= (probing ? kProbeSpan : kMaxBlockSpan)) && i < n) {
            // What did the match search buy?  Clean rows give few, long matches and a block several times smaller than its
            // literals alone would be; grain gives many short chance repeats that save a third of the block at five times the
            // encoding time (parse 120 ms + tokens 50 ms against 20 ms for a 4K frame).  Directory mode is bound by its encoder
            // threads on such content, so below a factor of two the following blocks are coded as literals — fifteen
            // of them, then a short block (64 KB) is parsed again to see what the rows look like now.
            const bool productive = write_block(bw, b, S.at(start), i - start, false) >= 2.0;
            if (!productive) lit_only = probing ? 15 : 7;
            probing = false;
            start = lit_start = i;
            S.block_done(i);
        }
    }
    write_block(bw, b, S.at(start), n - start, true);
trailer:
    bw.align();
    return (size_t)(bw.p - (out.data() + offset));
}

void put_adler(std::vector<uint8_t>& out, size_t offset, size_t& zn, uint32_t ad)
{
    uint8_t* o = out.data() + offset;
    o[zn++] = (uint8_t)(ad >> 24); o[zn++] = (uint8_t)(ad >> 16); o[zn++] = (uint8_t)(ad >> 8); o[zn++] = (uint8_t)ad;
}

}  // namespace

size_t fast_zlib_compress(const uint8_t* src, size_t n, std::vector<uint8_t>& out, size_t offset)
{
    if (n >= 0xffff0000u) {                      // positions are 32-bit below: hand a stream that long to zlib
        if (out.size() < offset + compressBound((uLong)n)) out.resize(offset + compressBound((uLong)n));
        uLongf cap = (uLongf)(out.size() - offset);
        return compress2(out.data() + offset, &cap, src, (uLong)n, 1) == Z_OK ? (size_t)cap : 0;
    }
    Contiguous S{src, n};
    size_t zn = compress_core(S, out, offset);
    put_adler(out, offset, zn, fast_adler32(1, src, n));
    return zn;
}

size_t fast_zlib_compress_rows(size_t rows, size_t row_bytes, const std::function<void(uint8_t* dst, size_t first_row, size_t n_rows)>& produce,
                               std::vector<uint8_t>& out, size_t offset)
{
    const size_t n = rows * row_bytes;
    if (rows == 0 || row_bytes == 0 || n / row_bytes != rows || n >= 0xffff0000u) return 0;
    static thread_local std::vector<uint8_t> window;
    uint32_t ad = 1;
    size_t next_row = 0;
    // (the Adler-32 of the stream is taken as the rows are produced: they are in L1 then)
    const std::function<void(uint8_t*, size_t)> fill = [&](uint8_t* dst, size_t k) {
        produce(dst, next_row, k);
        ad = fast_adler32(ad, dst, k * row_bytes);
        next_row += k;
    };
    RowWindow S(window, n, row_bytes, fill);
    size_t zn = compress_core(S, out, offset);
    put_adler(out, offset, zn, ad);
    return zn;
}

}  // namespace reve
