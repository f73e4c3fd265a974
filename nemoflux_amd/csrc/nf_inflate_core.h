// nf_inflate_core.h -- DEFLATE (RFC 1951) decoder inside a zlib (RFC 1950) wrapper, one WAVEFRONT per stream.
//
// Why it exists: real NEMO output is NetCDF-4 = HDF5 with float32 uo/vo stored as shuffled + deflated chunks (one per
// level in XIOS files); the reference reads them through netCDF4/xarray on the host (nemoflux/field.py:149), and host zlib
// is what bounds a file-backed pass.  Here the compressed chunks of a time step are copied to HBM
// as they are and every chunk is inflated by its own wavefront, hundreds at a time, straight into the staging slab the
// flux kernel reads.  Written from the two RFCs; no zlib code is used.
//
// Work inside the wavefront (the format is serial per stream, the parallelism is ACROSS streams):
//   input    all lanes keep the LDS input ring filled (coalesced 4-byte words from HBM)
//   decode   all lanes run the symbol decoder on UNIFORM values (scalar code): one LDS table lookup per symbol; a literal is
//            one byte store into the 32 KiB LDS window, a match is copied by the 64 lanes on the spot (overlapping copies by
//            the period rule) -- stream order, no queue
//   output   every 8 KiB the new bytes are flushed to HBM as whole 4-byte words and summed into the stream's Adler-32
//   tables   lane 0 assigns the canonical codes of a dynamic / fixed block, all lanes fill the lookup tables
//
// The same source compiles for the host (NFI_HOST: one "lane", no barriers) so that tests/ can run it against zlib's own
// output on the CPU; on the device it is driven by nf_inflate.hip.  Every loop is bounded by the input and output
// lengths: malformed data ends with an error code, never with a wild access.
#pragma once
#include <stdint.h>

#ifdef NFI_HOST
#define NFI_FN static inline
#define NFI_CONST static const
#define NFI_LANE 0
#define NFI_NLANE 1
#define NFI_SYNC() ((void)0)
#else
#define NFI_FN __device__ inline
#define NFI_CONST __constant__ static const
#define NFI_LANE ((int)threadIdx.x)
#define NFI_NLANE 64
// One stream = one wavefront = one workgroup, so "all lanes have done their LDS writes" needs no s_barrier and no wait for
// the LDS queue to drain: a wavefront's LDS instructions execute in issue order, a later read sees an earlier write of
// another lane.  What is needed is that the compiler keeps that order: a wavefront-scope fence + scheduling barrier (no
// instruction is emitted).  k_inflate is launched with exactly 64 lanes per workgroup.
#define NFI_SYNC()                                              \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)
#endif
#define NFI_FOR_LANES(i, n) for (int i = NFI_LANE; i < (int)(n); i += NFI_NLANE)
// The symbol decoder is serial and every lane would compute the same thing, so it runs as UNIFORM code: all lanes execute it,
// every value it reads from LDS is passed through readfirstlane, and the compiler keeps the whole bit-buffer arithmetic on
// the scalar unit (one instruction per cycle, native 64-bit shifts) instead of issuing 64-wide vector instructions for one
// useful lane -- 300 -> ~110 cycles per literal.
#ifdef NFI_HOST
#define NFI_UNI(x) ((uint32_t)(x))
#else
#define NFI_UNI(x) ((uint32_t)__builtin_amdgcn_readfirstlane((int)(x)))
#endif

enum {
    NFI_OK = 0,
    NFI_ERR_HEADER = 1,      // not a zlib stream (CMF/FLG), or a preset dictionary
    NFI_ERR_BLOCK = 2,       // reserved block type / stored-block length check
    NFI_ERR_CODES = 3,       // over-subscribed or unusable Huffman code set
    NFI_ERR_SYMBOL = 4,      // invalid code in the data
    NFI_ERR_DISTANCE = 5,    // match reaches before the start of the output
    NFI_ERR_OUTPUT = 6,      // more (or, at the end, fewer) bytes than the caller expects
    NFI_ERR_INPUT = 7,       // ran past the end of the compressed stream
    NFI_ERR_CHECKSUM = 8,    // Adler-32 of the output differs from the stream's trailer (RFC 1950)
};

constexpr int kNfiWindow = 32768;          // RFC 1951: distances up to 32 KiB
constexpr int kNfiRingWords = 512;         // input ring: two halves of 256 words (1 KiB each); a round reads < 100 words
constexpr int kNfiHalf = 256;
constexpr int kNfiQueue = 64;              // symbols decoded per round (between two looks at the input ring and the flush)
constexpr int kNfiLitBits = 10, kNfiDistBits = 8, kNfiClBits = 7;
constexpr uint32_t kNfiStoredRound = 512;  // bytes of a stored block moved per round (must stay inside one ring half)

template <int N> struct NfiHuffT {   // canonical code of one alphabet, for codes longer than the lookup table
    uint16_t count[16];              // number of codes of each length
    uint16_t symbol[N];              // symbols ordered by code
};
typedef NfiHuffT<288> NfiHuff;       // literal / length alphabet
typedef NfiHuffT<32> NfiHuffSmall;   // distance alphabet (30) and the code-length alphabet (19)

struct NfiCtx {           // lives in LDS (< 40 KiB, so that four fit a CU's 160 KiB): one per wavefront
    uint8_t window[kNfiWindow];
    uint32_t ring[kNfiRingWords];
    uint16_t lit_tab[1 << kNfiLitBits];    // (symbol << 4) | length, 0 = longer code
    uint16_t dist_tab[1 << kNfiDistBits];
    uint16_t cl_tab[1 << kNfiClBits];
    NfiHuff lit;
    NfiHuffSmall dist;
    uint8_t lens[320];                     // code lengths: 288 literal/length + 32 distance
    uint16_t code[320];                    // canonical code of every symbol (bit-reversed, as it appears in the stream)
    // state shared between the phases (written by lane 0, read by all after a barrier)
    uint64_t bitbuf;
    int32_t bitcnt;
    uint32_t word;          // next ring word to consume (index relative to the stream's first word)
    uint32_t loaded;        // ring words loaded so far
    uint32_t pos;           // output bytes produced
    uint32_t flushed;       // output bytes written to HBM
    int32_t state;          // 0 = need a block header, 1 = inside a Huffman block, 2 = inside a stored block, 3 = finished
    int32_t last;           // BFINAL of the current block
    uint32_t stored_left;
    int32_t err;
    int32_t nlit, ndist;
    uint32_t adler_a, adler_b;             // running Adler-32 of the flushed output
};

NFI_CONST uint8_t kNfiClOrder[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

// ---------------------------------------------------------------------------------------------- input ring (phase A)
// words: the stream's bytes seen as 4-byte words starting at the word that holds its first byte; nwords: how many of
// them may be read (the caller's buffer is padded, see nf_inflate.hip).  Loads the next half of the ring whenever the
// consumer has moved into the most recently loaded half.
NFI_FN void nfi_fill_ring(NfiCtx &c, const uint32_t *words, uint32_t nwords)
{
    while (c.loaded < c.word + kNfiHalf + 1) {       // uniform: c.word / c.loaded are read by all lanes after a barrier
        const uint32_t base = c.loaded;
        NFI_FOR_LANES(k, kNfiHalf) {
            const uint32_t w = base + (uint32_t)k;
            c.ring[w & (kNfiRingWords - 1)] = w < nwords ? words[w] : 0u;
        }
        NFI_SYNC();
        if (NFI_LANE == 0) c.loaded = base + kNfiHalf;
        NFI_SYNC();
    }
}

// ---------------------------------------------------------------------------------------------- bit reader (lane 0)
struct NfiBits {
    uint64_t buf;
    int cnt;
    uint32_t word;
};
NFI_FN void nfi_refill(const NfiCtx &c, NfiBits &b)
{
    if (b.cnt <= 32) {
        b.buf |= (uint64_t)c.ring[b.word & (kNfiRingWords - 1)] << b.cnt;
        b.cnt += 32;
        ++b.word;
    }
}
NFI_FN uint32_t nfi_take(NfiBits &b, int n)   // n <= 24, caller made sure cnt >= n
{
    const uint32_t v = (uint32_t)(b.buf & ((1u << n) - 1u));
    b.buf >>= n;
    b.cnt -= n;
    return v;
}

// decode one symbol: lookup table first, canonical walk for the codes that do not fit it (RFC 1951 3.2.2)
template <class H> NFI_FN int nfi_symbol(NfiBits &b, const uint16_t *tab, int tabbits, const H &h)
{
    const uint32_t e = NFI_UNI(tab[b.buf & ((1u << tabbits) - 1u)]);
    if (e & 15) {
        b.buf >>= (e & 15);
        b.cnt -= (int)(e & 15);
        return (int)(e >> 4);
    }
    int code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    for (int len = 1; len <= 15; ++len) {
        code |= (int)(bits & 1);
        bits >>= 1;
        const int count = (int)NFI_UNI(h.count[len]);
        if (code - count < first) {
            b.buf >>= len;
            b.cnt -= len;
            return (int)NFI_UNI(h.symbol[index + (code - first)]);
        }
        index += count;
        first += count;
        first <<= 1;
        code <<= 1;
    }
    return -1;
}

// ---------------------------------------------------------------------------------------------- table construction
// lens[0..n) -> canonical description (lane 0) + bit-reversed code of every symbol; returns "left" of the Kraft sum
// (0 = complete, > 0 = incomplete, < 0 = over-subscribed)
template <class H, class L> NFI_FN int nfi_canonical(H &h, const L *lens, uint16_t *code, int n)
{
    for (int l = 0; l <= 15; ++l) h.count[l] = 0;
    for (int s = 0; s < n; ++s) h.count[lens[s]]++;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= h.count[l];
        if (left < 0) return left;
    }
    uint16_t offs[16], next[16];
    offs[1] = 0;
    for (int l = 1; l < 15; ++l) offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
    int cc = 0;                                // RFC 1951 3.2.2: code = (code + bl_count[bits-1]) << 1, bl_count[0] = 0
    for (int l = 1; l <= 15; ++l) {
        cc = (cc + (l > 1 ? h.count[l - 1] : 0)) << 1;
        next[l] = (uint16_t)cc;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l) {
            h.symbol[offs[l]++] = (uint16_t)s;
            unsigned cw = next[l]++, r = 0;
            for (int k = 0; k < l; ++k) {        // the stream carries Huffman codes most significant bit first
                r = (r << 1) | (cw & 1u);
                cw >>= 1;
            }
            code[s] = (uint16_t)r;
        }
    }
    return left;
}

// all lanes: lookup table of `bits` bits from (lens, code)
NFI_FN void nfi_fill_table(uint16_t *tab, int bits, const uint8_t *lens, const uint16_t *code, int n)
{
    NFI_FOR_LANES(k, 1 << bits) tab[k] = 0;
    NFI_SYNC();
    NFI_FOR_LANES(s, n) {
        const int l = lens[s];
        if (l && l <= bits)
            for (unsigned k = code[s]; k < (1u << bits); k += (1u << l)) tab[k] = (uint16_t)((s << 4) | l);
    }
    NFI_SYNC();
}

// ---------------------------------------------------------------------------------------------- block header (phase B')
// lane 0 reads the header of the next block and, for Huffman blocks, leaves the code lengths in c.lens; returns through
// c.state / c.err.  The tables are then built by all lanes (nfi_build_tables).
NFI_FN void nfi_block_header(NfiCtx &c)
{
    NfiBits b{c.bitbuf, c.bitcnt, c.word};
    nfi_refill(c, b);
    c.last = (int)nfi_take(b, 1);
    const uint32_t type = nfi_take(b, 2);
    if (type == 0) {                       // stored: skip to the byte boundary, LEN, NLEN
        nfi_take(b, b.cnt & 7);
        nfi_refill(c, b);
        const uint32_t len = nfi_take(b, 16);
        nfi_refill(c, b);
        const uint32_t nlen = nfi_take(b, 16);
        if ((len ^ 0xffffu) != nlen) c.err = NFI_ERR_BLOCK;
        c.stored_left = len;
        c.state = 2;
    } else if (type == 1) {                // fixed codes (RFC 1951 3.2.6)
        for (int s = 0; s < 144; ++s) c.lens[s] = 8;
        for (int s = 144; s < 256; ++s) c.lens[s] = 9;
        for (int s = 256; s < 280; ++s) c.lens[s] = 7;
        for (int s = 280; s < 288; ++s) c.lens[s] = 8;
        for (int s = 0; s < 30; ++s) c.lens[288 + s] = 5;
        c.lens[318] = c.lens[319] = 0;
        c.nlit = 288;
        c.ndist = 30;
        c.state = 1;
    } else if (type == 2) {                // dynamic codes (3.2.7)
        nfi_refill(c, b);
        const int nlit = (int)nfi_take(b, 5) + 257, ndist = (int)nfi_take(b, 5) + 1, ncl = (int)nfi_take(b, 4) + 4;
        if (nlit > 286 || ndist > 30) {
            c.err = NFI_ERR_CODES;
        } else {
            uint16_t cl[19];
            for (int k = 0; k < 19; ++k) cl[k] = 0;
            for (int k = 0; k < ncl; ++k) {
                nfi_refill(c, b);
                cl[kNfiClOrder[k]] = (uint16_t)nfi_take(b, 3);
            }
            NfiHuffSmall &h = c.dist;      // scratch: the distance description is rebuilt right after
            uint16_t clcode[19];
            if (nfi_canonical(h, cl, clcode, 19) != 0 && !(h.count[0] == 18)) c.err = NFI_ERR_CODES;   // must be complete
            for (int k = 0; k < (1 << kNfiClBits); ++k) c.cl_tab[k] = 0;
            for (int s = 0; s < 19; ++s)
                if (cl[s])
                    for (unsigned k = clcode[s]; k < (1u << kNfiClBits); k += (1u << cl[s])) c.cl_tab[k] = (uint16_t)((s << 4) | cl[s]);
            int idx = 0;
            while (idx < nlit + ndist && !c.err) {
                nfi_refill(c, b);
                const int sym = nfi_symbol(b, c.cl_tab, kNfiClBits, h);
                if (sym < 0) {
                    c.err = NFI_ERR_SYMBOL;
                } else if (sym < 16) {
                    c.lens[idx < nlit ? idx : 288 + (idx - nlit)] = (uint8_t)sym;
                    ++idx;
                } else {
                    int rep, val = 0;
                    if (sym == 16) {
                        if (idx == 0) { c.err = NFI_ERR_CODES; break; }
                        const int prev = idx - 1;
                        val = c.lens[prev < nlit ? prev : 288 + (prev - nlit)];
                        rep = 3 + (int)nfi_take(b, 2);
                    } else if (sym == 17) {
                        rep = 3 + (int)nfi_take(b, 3);
                    } else {
                        rep = 11 + (int)nfi_take(b, 7);
                    }
                    if (idx + rep > nlit + ndist) { c.err = NFI_ERR_CODES; break; }
                    for (int r = 0; r < rep; ++r, ++idx) c.lens[idx < nlit ? idx : 288 + (idx - nlit)] = (uint8_t)val;
                }
            }
            for (int s = nlit; s < 288; ++s) c.lens[s] = 0;
            for (int s = ndist; s < 32; ++s) c.lens[288 + s] = 0;
            if (!c.err && c.lens[256] == 0) c.err = NFI_ERR_CODES;   // no end-of-block code
            c.nlit = nlit;
            c.ndist = ndist;
            c.state = 1;
        }
    } else {
        c.err = NFI_ERR_BLOCK;
    }
    c.bitbuf = b.buf;
    c.bitcnt = b.cnt;
    c.word = b.word;
}

NFI_FN void nfi_build_tables(NfiCtx &c)
{
    if (NFI_LANE == 0) {
        const int l1 = nfi_canonical(c.lit, c.lens, c.code, c.nlit);
        const int l2 = nfi_canonical(c.dist, c.lens + 288, c.code + 288, c.ndist);
        // over-subscribed sets are errors; an incomplete literal/length set too, unless it has a single code; the
        // distance set may be incomplete (one distance code, or none when the block holds literals only)
        if (l1 < 0 || l2 < 0 || (l1 > 0 && c.nlit - c.lit.count[0] != 1)) c.err = NFI_ERR_CODES;
    }
    NFI_SYNC();
    nfi_fill_table(c.lit_tab, kNfiLitBits, c.lens, c.code, c.nlit);
    nfi_fill_table(c.dist_tab, kNfiDistBits, c.lens + 288, c.code + 288, c.ndist);
}

// ---------------------------------------------------------------------------------------------- phase B: lane 0
// A match is copied 64 bytes per step by all lanes; an overlapping copy (dist < len) repeats the last dist bytes.
NFI_FN void nfi_copy_match(NfiCtx &c, uint32_t pos, uint32_t len, uint32_t dist)
{
    const float rcp = 1.0f / (float)dist;            // j < 258: the float quotient is exact to within one, fixed up below
    NFI_FOR_LANES(j, len) {
        uint32_t off = (uint32_t)j;                  // overlapping copy (dist < len): the bytes repeat with period dist
        if (off >= dist) {
            const uint32_t q = (uint32_t)((float)off * rcp);
            off -= q * dist;
            if ((int32_t)off < 0) off += dist;
            else if (off >= dist) off -= dist;
        }
        c.window[(pos + j) & (kNfiWindow - 1)] = c.window[(pos - dist + off) & (kNfiWindow - 1)];
    }
}

// decode up to kNfiQueue symbols of the current Huffman block straight into the LDS window.  Executed by ALL lanes on uniform
// values: a literal is one byte store (every lane writes the same byte to the same address), a match is copied by the 64
// lanes on the spot -- the symbols take effect in stream order, and a wavefront's LDS instructions execute in issue order, so
// no queue, no barrier and no reordering hazard is left.  A wavefront issues one instruction every four cycles at best, so
// what this loop costs is its instruction count: the run of short-coded literals -- most symbols of real data -- is a loop of
// its own with one table lookup, one byte store and a handful of scalar instructions per byte.
NFI_FN void nfi_decode_round(NfiCtx &c, uint32_t out_len)
{
    NfiBits b;      // executed by ALL lanes on uniform values (see NFI_UNI): scalar code
    b.buf = ((uint64_t)NFI_UNI((uint32_t)(c.bitbuf >> 32)) << 32) | NFI_UNI((uint32_t)c.bitbuf);
    b.cnt = (int)NFI_UNI(c.bitcnt);
    b.word = NFI_UNI(c.word);
    const int last = (int)NFI_UNI(c.last);
    uint32_t nextw = NFI_UNI(c.ring[b.word & (kNfiRingWords - 1)]);
#define NFI_REFILL()                                                     \
    if (b.cnt <= 32) {                                                   \
        b.buf |= (uint64_t)nextw << b.cnt;                               \
        b.cnt += 32;                                                     \
        ++b.word;                                                        \
        nextw = NFI_UNI(c.ring[b.word & (kNfiRingWords - 1)]);           \
    }
    uint32_t pos = NFI_UNI(c.pos);
    uint32_t nq = 0;                              // symbols of this round
    int err = 0, state = 1;
    constexpr uint32_t kMask = (1u << kNfiLitBits) - 1u;
    for (;;) {
        // ---- run of short-coded literals.  A table entry is (symbol << 4) | length, 0 for a code longer than the table:
        // entries 1 .. 4095 are exactly the literals 0 .. 255.
        uint32_t rem = kNfiQueue - nq;
        if (out_len - pos < rem) rem = out_len - pos;
        uint32_t e = 0;
        bool have = false;                        // e holds the entry of the symbol at the head of the bit buffer
        while (rem) {
            NFI_REFILL();
            e = NFI_UNI(c.lit_tab[(uint32_t)b.buf & kMask]);
            have = true;
            if (e - 1u >= 4095u) break;
            b.buf >>= (e & 15);
            b.cnt -= (int)(e & 15);
            c.window[pos & (kNfiWindow - 1)] = (uint8_t)(e >> 4);
            ++nq;
            ++pos;
            --rem;
            have = false;
        }
        if (nq == kNfiQueue) break;
        // ---- any other symbol
        if (!have) {                              // the output is full: only the end-of-block code may follow
            NFI_REFILL();
            e = NFI_UNI(c.lit_tab[(uint32_t)b.buf & kMask]);
        }
        int sym;
        if (e & 15) {
            b.buf >>= (e & 15);
            b.cnt -= (int)(e & 15);
            sym = (int)(e >> 4);
        } else {
            sym = nfi_symbol(b, c.lit_tab, kNfiLitBits, c.lit);      // takes the canonical walk (the entry is 0)
            if (sym < 0) { err = NFI_ERR_SYMBOL; break; }
        }
        if (sym < 256) {                          // a literal with a long code, or one that does not fit any more
            if (pos >= out_len) { err = NFI_ERR_OUTPUT; break; }
            c.window[pos & (kNfiWindow - 1)] = (uint8_t)sym;
            ++nq;
            ++pos;
        } else if (sym == 256) {
            state = last ? 3 : 0;
            break;
        } else {
            const int li = sym - 257;
            if (li >= 29) { err = NFI_ERR_SYMBOL; break; }
            // base and extra bits of the length / distance codes (RFC 1951 3.2.5) by arithmetic: a table in memory would
            // cost a scalar-memory round trip per symbol
            const int lx = li < 8 ? 0 : (li - 4) >> 2;
            const uint32_t lbase = li < 8 ? 3u + (uint32_t)li : (li == 28 ? 258u : 3u + ((4u + (uint32_t)(li & 3)) << lx));
            const uint32_t len = lbase + nfi_take(b, li == 28 ? 0 : lx);
            NFI_REFILL();
            const int ds = nfi_symbol(b, c.dist_tab, kNfiDistBits, c.dist);
            if (ds < 0 || ds >= 30) { err = NFI_ERR_SYMBOL; break; }
            const int dx = ds < 4 ? 0 : (ds - 2) >> 1;
            const uint32_t dbase = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + (uint32_t)(ds & 1)) << dx);
            const uint32_t dist = dbase + nfi_take(b, dx);
            if (dist > pos) { err = NFI_ERR_DISTANCE; break; }
            if (pos + len > out_len) { err = NFI_ERR_OUTPUT; break; }
            nfi_copy_match(c, pos, len, dist);          // len in 3 .. 258, dist in 1 .. 32768
            ++nq;
            pos += len;
        }
    }
#undef NFI_REFILL
    if (err) c.err = err;
    c.state = state;
    c.pos = pos;
    c.bitbuf = b.buf;
    c.bitcnt = b.cnt;
    c.word = b.word;
}

// ---------------------------------------------------------------------------------------------- phase C: all lanes
// bytes of a stored block: straight from the input ring into the window (all lanes), at most one ring half per call
NFI_FN void nfi_stored_round(NfiCtx &c, uint32_t out_len)
{
    // the bit buffer holds whole bytes here (the header aligned it); give them back to the ring position
    uint32_t n = c.stored_left < kNfiStoredRound ? c.stored_left : kNfiStoredRound;
    if (c.pos + n > out_len) {
        if (NFI_LANE == 0) c.err = NFI_ERR_OUTPUT;
        NFI_SYNC();
        return;
    }
    const uint32_t have = (uint32_t)c.bitcnt >> 3;                       // bytes still in the bit buffer
    const uint64_t buf = c.bitbuf;
    const uint32_t byte0 = c.word * 4u;                                   // ring byte that follows the bit buffer
    const uint32_t pos = c.pos;
    NFI_FOR_LANES(j, n) {
        uint8_t v;
        if ((uint32_t)j < have) v = (uint8_t)(buf >> (8 * j));
        else {
            const uint32_t bi = byte0 + ((uint32_t)j - have);
            v = (uint8_t)(c.ring[(bi >> 2) & (kNfiRingWords - 1)] >> (8 * (bi & 3)));
        }
        c.window[(pos + j) & (kNfiWindow - 1)] = v;
    }
    NFI_SYNC();
    if (NFI_LANE == 0) {
        if (n <= have) {
            c.bitbuf = have == 8 && n == 8 ? 0 : (buf >> (8 * n));
            c.bitcnt -= 8 * (int)n;
        } else {
            const uint32_t bi = byte0 + (n - have);                       // first unread byte
            c.word = bi >> 2;
            const uint32_t sub = bi & 3;
            if (sub) {                                                    // keep the rest of that word in the bit buffer
                c.bitbuf = (uint64_t)(c.ring[c.word & (kNfiRingWords - 1)] >> (8 * sub));
                c.bitcnt = 32 - 8 * (int)sub;
                ++c.word;
            } else {
                c.bitbuf = 0;
                c.bitcnt = 0;
            }
        }
        c.pos = pos + n;
        c.stored_left -= n;
        if (c.stored_left == 0) c.state = c.last ? 3 : 0;
    }
    NFI_SYNC();
}

// write the finished part of the window to the output: whole 4-byte words (dst 4-byte aligned), the tail at the end;
// the Adler-32 of the stream (RFC 1950) is carried along: for n new bytes b_0..b_{n-1}, a += sum b_i and
// b += n*a_old + sum (n-i) b_i, both modulo 65521 -- per-lane partial sums, added up by lane 0
constexpr uint32_t kNfiFlushBytes = 8192;   // + one round's output (<= 64 x 258) stays below the 32 KiB the ring holds
NFI_FN void nfi_flush(NfiCtx &c, uint8_t *dst, bool final)
{
    if (!final && c.pos - c.flushed < kNfiFlushBytes) return;     // uniform: both are read after a barrier
    const uint32_t from = c.flushed, upto = final ? c.pos : (c.pos & ~3u);
    const uint32_t n = upto - from;
    uint32_t pa = 0;
    uint64_t pb = 0;                            // 64 bits: the host build runs all of a flush (up to 16.5 KiB) on one "lane"
    if (((uintptr_t)dst & 3u) == 0) {
        const uint32_t nwords = n >> 2;
        uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + from);
        NFI_FOR_LANES(w, nwords) {
            const uint32_t i0 = 4u * (uint32_t)w, p = (from + i0) & (kNfiWindow - 1);
            const uint32_t b0 = c.window[p], b1 = c.window[p + 1], b2 = c.window[p + 2], b3 = c.window[p + 3];
            d32[w] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
            pa += b0 + b1 + b2 + b3;
            pb += (uint64_t)((n - i0) * b0 + (n - i0 - 1) * b1) + (uint64_t)((n - i0 - 2) * b2 + (n - i0 - 3) * b3);
        }
        const uint32_t done = 4u * nwords;
        NFI_FOR_LANES(j, n - done) {
            const uint32_t v = c.window[(from + done + j) & (kNfiWindow - 1)];
            dst[from + done + j] = (uint8_t)v;
            pa += v;
            pb += (n - done - (uint32_t)j) * v;
        }
    } else {
        NFI_FOR_LANES(j, n) {
            const uint32_t v = c.window[(from + j) & (kNfiWindow - 1)];
            dst[from + j] = (uint8_t)v;
            pa += v;
            pb += (n - (uint32_t)j) * v;
        }
    }
    uint32_t sa = pa, sb = (uint32_t)(pb % 65521u);
#ifndef NFI_HOST
    for (int o = 32; o > 0; o >>= 1) {          // wavefront sums (64 x 65520 and 64 x 16.5 KiB x 255 fit 32 bits)
        sa += __shfl_xor(sa, o, 64);
        sb += __shfl_xor(sb, o, 64);
    }
#endif
    if (NFI_LANE == 0) {
        const uint32_t a_old = c.adler_a;
        c.adler_a = (a_old + sa) % 65521u;
        c.adler_b = (uint32_t)((c.adler_b + (uint64_t)n * a_old + sb) % 65521u);
        c.flushed = upto;
    }
    NFI_SYNC();
}

// ---------------------------------------------------------------------------------------------- one whole stream
// src: first byte of the zlib stream; readable: bytes that may be read starting at (src rounded down to 4) -- at least the
// stream, the tail is never interpreted beyond in_len + 8.  Returns NFI_OK and exactly out_len bytes in dst, or an error.
NFI_FN int nfi_inflate_stream(NfiCtx &c, const uint8_t *src, uint32_t in_len, uint32_t readable, uint8_t *dst,
                              uint32_t out_len)
{
    const uint32_t skip = (uint32_t)((uintptr_t)src & 3u);
    const uint32_t *words = reinterpret_cast<const uint32_t *>(src - skip);
    const uint32_t nwords = readable >> 2;
    if (NFI_LANE == 0) {
        c.bitbuf = 0;
        c.bitcnt = 0;
        c.word = 0;
        c.loaded = 0;
        c.pos = 0;
        c.flushed = 0;
        c.state = 0;
        c.last = 0;
        c.err = in_len < 6 ? NFI_ERR_INPUT : NFI_OK;
        c.stored_left = 0;
        c.adler_a = 1;
        c.adler_b = 0;
    }
    NFI_SYNC();
    nfi_fill_ring(c, words, nwords);
    if (NFI_LANE == 0 && !c.err) {                       // zlib header (RFC 1950): CM = 8, no dictionary, check bits
        NfiBits b{0, 0, 0};
        nfi_refill(c, b);
        nfi_take(b, 8 * (int)skip);                      // bytes of the first word that precede the stream
        nfi_refill(c, b);
        const uint32_t cmf = nfi_take(b, 8), flg = nfi_take(b, 8);
        if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 32)) c.err = NFI_ERR_HEADER;
        c.bitbuf = b.buf;
        c.bitcnt = b.cnt;
        c.word = b.word;
    }
    NFI_SYNC();
    const uint32_t word_limit = ((skip + in_len + 3u) >> 2) + 3u;     // consuming beyond this = reading past the stream
    while (!c.err && c.state != 3) {                                  // uniform: state / err are read after barriers
        nfi_fill_ring(c, words, nwords);
        if (c.state == 0) {
            if (NFI_LANE == 0) nfi_block_header(c);
            NFI_SYNC();
            if (!c.err && c.state == 1) nfi_build_tables(c);
        } else if (c.state == 1) {
            nfi_decode_round(c, out_len);          // all lanes, uniform
            NFI_SYNC();
            nfi_flush(c, dst, false);
        } else {
            nfi_stored_round(c, out_len);
            nfi_flush(c, dst, false);
        }
        if (NFI_LANE == 0 && c.word > word_limit) c.err = NFI_ERR_INPUT;
        NFI_SYNC();
    }
    if (!c.err) nfi_flush(c, dst, true);
    if (NFI_LANE == 0 && !c.err) {
        if (c.pos != out_len) c.err = NFI_ERR_OUTPUT;
        else {                                           // trailer: Adler-32 of the uncompressed data, big-endian
            const uint8_t *t = src + in_len - 4;
            const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
            if (want != ((c.adler_b << 16) | c.adler_a)) c.err = NFI_ERR_CHECKSUM;
        }
    }
    NFI_SYNC();
    return c.err;
}
