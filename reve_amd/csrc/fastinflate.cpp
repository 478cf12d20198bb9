#include "fastinflate.h"

#include <cstring>
#include <memory>

#include "fastdeflate.h"      // fast_adler32

// Written from RFC 1950 / RFC 1951 for the one job directory mode has: turn the IDAT stream of a decoded video frame (grain: few
// matches, mostly literals of 4-6 bits) into scanlines faster than zlib's inflate does (3.4 -> ~1.2 ns per output byte).  The
// means are the usual ones of a table-driven decoder: a 64-bit bit buffer refilled by one unaligned load, a 12-bit first-level
// table for literal / length codes (8-bit for distances) whose entries carry everything a symbol needs — code length, kind,
// extra-bit count, base value — second-level tables for the rare longer codes, up to three literals per refill, and match copies
// in 8-byte words.  Input is untrusted (anything may sit in tmp_frames/): every table index is masked, every distance is checked
// against what has been written, every copy against what is left, and a stream that decodes to anything but exactly the
// expected size is an error.  tests: tests/test_abi.py (against zlib on every kind of stream, truncations, bit flips),
// tests/test_sanitizers.py (the same under AddressSanitizer).
namespace reve {
namespace {

constexpr int kLitBits = 12, kDistBits = 8, kMaxCodeLen = 15;      // (first-level bits: 12 against 11 is 25 % less time on literal-only streams — more pairs)
constexpr int kLitSub = 1 << (kMaxCodeLen - kLitBits), kDistSub = 1 << (kMaxCodeLen - kDistBits);      // entries of a second-level table
// entry: bits 0-7 code length to consume (second-level pointer: index bits of that table), 8-11 kind, 12-15 extra bits, 16-31 value
enum : uint32_t { K_LIT = 0, K_LEN = 1, K_EOB = 2, K_SUB = 3, K_BAD = 4, K_DIST = 5 };
constexpr uint32_t entry(uint32_t len, uint32_t kind, uint32_t extra, uint32_t value) { return len | (kind << 8) | (extra << 12) | (value << 16); }
constexpr uint32_t kBad = entry(1, K_BAD, 0, 0);

constexpr uint16_t kLenBase[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
constexpr uint8_t kLenExtra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
constexpr uint16_t kDistBase[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
constexpr uint8_t kDistExtra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
constexpr uint8_t kClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

struct Tables {
    uint32_t lit[(1 << kLitBits) + 288 * kLitSub];
    uint32_t dist[(1 << kDistBits) + 32 * kDistSub];
};

inline uint32_t reverse_bits(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

// Canonical Huffman decoding table from code lengths (RFC 1951 §3.2.2).  `what`: 0 literal / length, 1 distance, 2 code lengths.
// Over-subscribed sets are errors; incomplete ones too, except the single one-bit code zlib also lets through (a block with one
// distance code) and the empty distance set of a block without matches.  Unused slots decode to K_BAD.
bool build_table(const uint8_t* lens, int n, int what, uint32_t* tab, int prim_bits, int sub_size)
{
    int count[kMaxCodeLen + 1] = {};
    for (int s = 0; s < n; ++s) count[lens[s]]++;
    count[0] = 0;
    int left = 1, maxlen = 0;
    for (int l = 1; l <= kMaxCodeLen; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return false;                                   // over-subscribed
        if (count[l]) maxlen = l;
    }
    if (left > 0 && !(maxlen <= 1 && what != 2 && (what == 1 || maxlen == 1))) return false;     // incomplete
    const int prim = 1 << prim_bits;
    for (int i = 0; i < prim; ++i) tab[i] = kBad;
    uint32_t next_code[kMaxCodeLen + 2] = {};
    {
        uint32_t c = 0;
        for (int l = 1; l <= kMaxCodeLen; ++l) { c = (c + (uint32_t)count[l - 1]) << 1; next_code[l] = c; }
    }
    int n_sub = 0;
    for (int s = 0; s < n; ++s) {
        const int len = lens[s];
        if (!len) continue;
        const uint32_t rev = reverse_bits(next_code[len]++, len);
        uint32_t e;
        if (what == 0) {
            if (s < 256) e = entry((uint32_t)len, K_LIT, 1, (uint32_t)s);
            else if (s == 256) e = entry((uint32_t)len, K_EOB, 0, 0);
            else if (s < 286) e = entry((uint32_t)len, K_LEN, kLenExtra[s - 257], kLenBase[s - 257]);
            else e = entry((uint32_t)len, K_BAD, 0, 0);                // 286, 287: in the fixed code, never valid in data
        } else if (what == 1) {
            e = s < 30 ? entry((uint32_t)len, K_DIST, kDistExtra[s], kDistBase[s]) : entry((uint32_t)len, K_BAD, 0, 0);
        } else {
            e = entry((uint32_t)len, K_LIT, 0, (uint32_t)s);
        }
        if (len <= prim_bits) {
            for (uint32_t i = rev; i < (uint32_t)prim; i += 1u << len) tab[i] = e;
        } else {
            // second level: one table of sub_size entries per distinct first-level prefix, indexed by the bits after the prefix
            const uint32_t pre = rev & (uint32_t)(prim - 1);
            uint32_t at;
            if (((tab[pre] >> 8) & 15) == K_SUB) {
                at = tab[pre] >> 16;
            } else {
                at = (uint32_t)(prim + n_sub * sub_size);
                ++n_sub;
                for (int i = 0; i < sub_size; ++i) tab[at + i] = kBad;
                tab[pre] = entry((uint32_t)(kMaxCodeLen - prim_bits), K_SUB, 0, at);
            }
            const uint32_t hi = rev >> prim_bits;                         // the code's remaining len - prim_bits bits
            e = (e & ~0xffu) | (uint32_t)(len - prim_bits);               // (consumed after the prefix)
            for (uint32_t i = hi; i < (uint32_t)sub_size; i += 1u << (len - prim_bits)) tab[at + i] = e;
        }
    }
    if (what == 0) {
        // Two literals per lookup: where a first-level index holds a literal's code AND, in the bits above it, the complete code of a
        // second literal, the entry yields both (count 2 in the `extra` field, the second byte in the value's high half).  Decoded
        // video is mostly literals of 3-6 bits, and a Huffman decoder is one dependency chain — table load, shift, mask, ~7 cycles
        // per symbol — so halving the lookups is what halves the time.  (Entries are replicated over their unused index bits: an entry
        // whose length fits the bits that ARE known does not depend on the unknown ones.)
        uint32_t single[1 << kLitBits];
        std::memcpy(single, tab, sizeof(uint32_t) * (size_t)prim);
        for (int i = 0; i < prim; ++i) {
            const uint32_t e1 = single[i];
            if (((e1 >> 8) & 15) != K_LIT) continue;
            const int l1 = (int)(e1 & 0xff);
            const uint32_t e2 = single[(uint32_t)i >> l1];         // (the unknown upper bits read as zero)
            const int l2 = (int)(e2 & 0xff);
            if (((e2 >> 8) & 15) != K_LIT || l1 + l2 > prim_bits) continue;
            tab[i] = entry((uint32_t)(l1 + l2), K_LIT, 2, (e1 >> 16) | ((e2 >> 16) << 8));
        }
    }
    return true;
}

inline uint64_t load64(const uint8_t* p) { uint64_t v; std::memcpy(&v, p, 8); return v; }

struct Reader {
    const uint8_t* ip;
    const uint8_t* end;
    uint64_t buf = 0;
    int cnt = 0;                   // valid bits in buf
    bool overrun = false;          // bits were asked for beyond the end of the input
    inline void refill_fast()      // needs ip + 8 <= end; afterwards cnt >= 56
    {
        buf |= load64(ip) << cnt;
        ip += (63 - cnt) >> 3;
        cnt |= 56;
    }
    inline void refill()           // anywhere; bytes beyond the end read as zero (and are flagged when consumed)
    {
        if (end - ip >= 8) { refill_fast(); return; }
        while (cnt <= 56 && ip < end) { buf |= (uint64_t)*ip++ << cnt; cnt += 8; }
    }
    inline uint32_t peek(int n) const { return (uint32_t)(buf & (((uint64_t)1 << n) - 1)); }
    inline void drop(int n)
    {
        if (n > cnt) { overrun = true; n = cnt; }
        buf >>= n;
        cnt -= n;
    }
    inline uint32_t take(int n) { const uint32_t v = peek(n); drop(n); return v; }
};

const Tables& fixed_tables()
{
    static const Tables* t = [] {
        Tables* x = new Tables;
        uint8_t l[288], d[32];
        for (int s = 0; s < 288; ++s) l[s] = s < 144 ? 8 : s < 256 ? 9 : s < 280 ? 7 : 8;
        for (int s = 0; s < 32; ++s) d[s] = 5;
        (void)build_table(l, 288, 0, x->lit, kLitBits, kLitSub);
        (void)build_table(d, 32, 1, x->dist, kDistBits, kDistSub);
        return x;
    }();
    return *t;
}

// the symbols of one block; returns "" at its end-of-block code
const char* inflate_block(Reader& r, const Tables& t, uint8_t* const out, uint8_t*& op, uint8_t* const out_end)
{
    for (;;) {
        // ---- fast loop: 8 input bytes and 274 output bytes (a longest match + a word of overshoot + 2 x 2 literals + 1) are there.  The
        // reader's state lives in locals meanwhile: the output stores go through a byte pointer, which may alias anything reachable
        // through a reference, and would force its fields to be re-read after every literal.
        if (r.end - r.ip >= 8 && out_end - op >= 274) {
            const uint8_t* ip = r.ip;
            const uint8_t* const in_end = r.end;
            uint64_t buf = r.buf;
            int cnt = r.cnt;
            uint8_t* o = op;
            const char* err = nullptr;
            bool eob = false;
#define REVE_REFILL() do { buf |= load64(ip) << cnt; ip += (63 - cnt) >> 3; cnt |= 56; } while (0)
            while (in_end - ip >= 8 && out_end - o >= 274) {
                REVE_REFILL();
                uint32_t e = t.lit[buf & ((1u << kLitBits) - 1)];
                int lits = 0;
                // up to three literals per refill (3 x 15 bits); the first non-literal falls through with its entry in `e`
                while ((e & 0xf00) == (K_LIT << 8)) {
                    buf >>= e & 0xff; cnt -= (int)(e & 0xff);
                    const uint16_t two = (uint16_t)(e >> 16);          // one or two literals (the second byte is overwritten if one)
                    std::memcpy(o, &two, 2);
                    o += (e >> 12) & 15;
                    if (++lits == 3) break;
                    e = t.lit[buf & ((1u << kLitBits) - 1)];
                }
                if (lits == 3) continue;
                if (lits && cnt < 48) {                              // a length + distance needs up to 48 bits
                    if (in_end - ip < 8) break;
                    REVE_REFILL();
                }
                uint32_t kind = (e >> 8) & 15;
                if (kind == K_SUB) {
                    buf >>= kLitBits; cnt -= kLitBits;
                    e = t.lit[(e >> 16) + (buf & (kLitSub - 1))];
                    kind = (e >> 8) & 15;
                    if (kind == K_LIT) { buf >>= e & 0xff; cnt -= (int)(e & 0xff); *o++ = (uint8_t)(e >> 16); continue; }      // (second level: single literals)
                }
                if (kind == K_LEN) {
                    buf >>= e & 0xff; cnt -= (int)(e & 0xff);
                    const int xb = (int)((e >> 12) & 15);
                    const size_t len = (e >> 16) + (buf & ((1u << xb) - 1));
                    buf >>= xb; cnt -= xb;
                    uint32_t d = t.dist[buf & ((1u << kDistBits) - 1)];
                    if (((d >> 8) & 15) == K_SUB) {
                        buf >>= kDistBits; cnt -= kDistBits;
                        d = t.dist[(d >> 16) + (buf & (kDistSub - 1))];
                    }
                    if (((d >> 8) & 15) != K_DIST) { err = "invalid distance code"; break; }
                    buf >>= d & 0xff; cnt -= (int)(d & 0xff);
                    const int db = (int)((d >> 12) & 15);
                    const size_t dist = (d >> 16) + (buf & ((1u << db) - 1));
                    buf >>= db; cnt -= db;
                    if (cnt < 0) { err = "deflate stream ends inside a symbol"; break; }
                    if (dist > (size_t)(o - out)) { err = "distance reaches before the start of the data"; break; }
                    const uint8_t* s = o - dist;
                    uint8_t* const e_o = o + len;
                    if (dist >= 8) {
                        do { std::memcpy(o, s, 8); o += 8; s += 8; } while (o < e_o);
                    } else if (dist == 1) {
                        const uint64_t v = 0x0101010101010101ull * *s;
                        do { std::memcpy(o, &v, 8); o += 8; } while (o < e_o);
                    } else {
                        // a period shorter than a word: the first bytes one by one, then whole words at a multiple of the period
                        const size_t big = dist * ((7 + dist) / dist);       // 8..14
                        size_t k = 0;
                        for (; k < big && o + k < e_o; ++k) o[k] = s[k];
                        uint8_t* q = o + k;
                        while (q < e_o) { std::memcpy(q, q - big, 8); q += 8; }
                    }
                    o = e_o;
                    continue;
                }
                if (kind == K_EOB) {
                    buf >>= e & 0xff; cnt -= (int)(e & 0xff);
                    if (cnt < 0) err = "deflate stream ends inside a symbol";
                    eob = true;
                    break;
                }
                err = "invalid literal/length code";
                break;
            }
#undef REVE_REFILL
            r.ip = ip; r.buf = buf; r.cnt = cnt;
            op = o;
            if (err) return err;
            if (eob) return "";
        }
        // ---- careful path: one symbol, everything checked (the stream's and the buffer's last bytes)
        r.refill();
        uint32_t e = t.lit[r.peek(kLitBits)];
        if (((e >> 8) & 15) == K_SUB) { r.drop(kLitBits); e = t.lit[(e >> 16) + r.peek(kMaxCodeLen - kLitBits)]; }
        const uint32_t kind = (e >> 8) & 15;
        r.drop((int)(e & 0xff));
        if (r.overrun) return "deflate stream ends inside a symbol";
        if (kind == K_LIT) {
            const size_t k = (e >> 12) & 15;                     // one or two literals
            if ((size_t)(out_end - op) < k) return "more data than the image holds";
            *op++ = (uint8_t)(e >> 16);
            if (k == 2) *op++ = (uint8_t)(e >> 24);
        } else if (kind == K_EOB) {
            return "";
        } else if (kind == K_LEN) {
            size_t len = (e >> 16) + r.take((int)((e >> 12) & 15));
            r.refill();
            uint32_t d = t.dist[r.peek(kDistBits)];
            if (((d >> 8) & 15) == K_SUB) { r.drop(kDistBits); d = t.dist[(d >> 16) + r.peek(kMaxCodeLen - kDistBits)]; }
            if (((d >> 8) & 15) != K_DIST) return "invalid distance code";
            r.drop((int)(d & 0xff));
            const size_t dist = (d >> 16) + r.take((int)((d >> 12) & 15));
            if (r.overrun) return "deflate stream ends inside a symbol";
            if (dist > (size_t)(op - out)) return "distance reaches before the start of the data";
            if (len > (size_t)(out_end - op)) return "more data than the image holds";
            for (const uint8_t* s = op - dist; len; --len) *op++ = *s++;
        } else {
            return "invalid literal/length code";
        }
    }
}

const char* inflate_raw(Reader& r, uint8_t* const out, size_t out_len)
{
    uint8_t* op = out;
    uint8_t* const out_end = out + out_len;
    static thread_local std::unique_ptr<Tables> dyn;
    for (;;) {
        r.refill();
        const uint32_t final = r.take(1), type = r.take(2);
        if (r.overrun) return "deflate stream ends inside a block header";
        if (type == 0) {
            // stored: back to a byte boundary — the reader has fetched ahead, its whole bytes go back to the input
            r.drop(r.cnt & 7);
            r.ip -= r.cnt >> 3;
            r.buf = 0; r.cnt = 0;
            if (r.end - r.ip < 4) return "deflate stream ends inside a stored block";
            const uint32_t len = r.ip[0] | ((uint32_t)r.ip[1] << 8), nlen = r.ip[2] | ((uint32_t)r.ip[3] << 8);
            r.ip += 4;
            if ((len ^ 0xffff) != nlen) return "stored block length check failed";
            if (len > (size_t)(r.end - r.ip)) return "deflate stream ends inside a stored block";
            if (len > (size_t)(out_end - op)) return "more data than the image holds";
            if (len) std::memcpy(op, r.ip, len);
            op += len;
            r.ip += len;
        } else if (type == 1) {
            if (const char* e = inflate_block(r, fixed_tables(), out, op, out_end); e[0]) return e;
        } else if (type == 2) {
            const uint32_t hlit = r.take(5) + 257, hdist = r.take(5) + 1, hclen = r.take(4) + 4;
            if (hlit > 286 || hdist > 30) return "too many length or distance codes";
            uint8_t cl[19] = {};
            for (uint32_t i = 0; i < hclen; ++i) { r.refill(); cl[kClOrder[i]] = (uint8_t)r.take(3); }
            if (r.overrun) return "deflate stream ends inside a block header";
            uint32_t cltab[1 << 7];
            if (!build_table(cl, 19, 2, cltab, 7, 1)) return "invalid code-length code";
            uint8_t lens[288 + 32] = {};
            for (uint32_t i = 0; i < hlit + hdist;) {
                r.refill();
                const uint32_t e = cltab[r.peek(7)];
                if (((e >> 8) & 15) != K_LIT) return "invalid code-length symbol";
                r.drop((int)(e & 0xff));
                const uint32_t sym = e >> 16;
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                uint32_t rep, v = 0;
                if (sym == 16) {
                    if (i == 0) return "code-length repeat with nothing to repeat";
                    v = lens[i - 1];
                    rep = 3 + r.take(2);
                } else if (sym == 17) rep = 3 + r.take(3);
                else rep = 11 + r.take(7);
                if (i + rep > hlit + hdist) return "code-length repeat runs past the end";
                while (rep--) lens[i++] = (uint8_t)v;
            }
            if (r.overrun) return "deflate stream ends inside a block header";
            if (!lens[256]) return "block without an end-of-block code";
            if (!dyn) dyn.reset(new Tables);             // 60 KB per decoding thread, kept
            uint8_t dl[32] = {};
            std::memcpy(dl, lens + hlit, hdist);
            if (!build_table(lens, (int)hlit, 0, dyn->lit, kLitBits, kLitSub)) return "invalid literal/length code set";
            if (!build_table(dl, (int)hdist, 1, dyn->dist, kDistBits, kDistSub)) return "invalid distance code set";
            if (const char* e = inflate_block(r, *dyn, out, op, out_end); e[0]) return e;
        } else {
            return "invalid block type";
        }
        if (final) break;
    }
    if (op != out_end) return "less data than the image holds";
    return "";
}

}  // namespace

std::string fast_zlib_uncompress(const uint8_t* in, size_t in_len, uint8_t* out, size_t out_len)
{
    if (in_len < 6) return "zlib stream too short";
    if ((in[0] & 15) != 8 || (in[0] >> 4) > 7 || ((in[0] << 8) | in[1]) % 31 != 0) return "not a zlib stream";
    if (in[1] & 0x20) return "zlib stream wants a preset dictionary";
    Reader r;
    r.ip = in + 2;
    r.end = in + in_len;
    if (const char* e = inflate_raw(r, out, out_len); e[0]) return e;
    // the Adler-32 of the data follows the last block, on a byte boundary
    r.drop(r.cnt & 7);
    r.ip -= r.cnt >> 3;
    if (r.end - r.ip < 4) return "zlib stream ends before its checksum";
    const uint32_t want = ((uint32_t)r.ip[0] << 24) | ((uint32_t)r.ip[1] << 16) | ((uint32_t)r.ip[2] << 8) | r.ip[3];
    if (fast_adler32(1, out, out_len) != want) return "zlib checksum mismatch";
    return "";
}

}  // namespace reve
