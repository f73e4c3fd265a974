// nf_inflate_core.h -- DEFLATE (RFC 1951) decoder inside a zlib (RFC 1950) wrapper, one WAVEFRONT per stream.
//
// Why it exists: real NEMO output is NetCDF-4 = HDF5 with float32 uo/vo stored as shuffled + deflated chunks (one per
// level in XIOS files); the reference reads them through netCDF4/xarray on the host (nemoflux/field.py:149), and host zlib
// is what bounds a file-backed pass ("Ingest").  Here the compressed chunks of a time step are copied to HBM
// as they are and every chunk is inflated by its own wavefront, a thousand at a time, into the staging slab the flux kernel
// reads.  Written from the two RFCs; no zlib code is used.
//
// Work inside the wavefront (the format is serial per stream, the parallelism is ACROSS streams):
//   input    all lanes keep the LDS input ring filled (coalesced 4-byte words from HBM)
//   decode   all lanes run the symbol decoder on UNIFORM values (scalar code).  The decoder is a chain of dependent table
//            lookups, so what it costs is LOOKUPS PER BYTE and what sits between two lookups:
//              * a table entry carries everything about its symbol -- one literal or TWO (when both codes fit the index),
//                or a length's base + extra-bit count, or a distance's -- so no symbol needs arithmetic or a second
//                dependent read to be understood;
//              * the device's hot loop is hand-written ISA with the tables held in REGISTERS (nfi_decode_round_asm: a lookup
//                is VGPR index mode + v_readlane, no LDS round trip); the portable form (nfi_decode_round_cxx: the host build,
//                and the rare cases on the device) requests the entry of the NEXT symbol before the current one takes effect;
//              * a literal is one byte store into the 32 KiB LDS window, a match is copied by the 64 lanes on the spot
//                (overlapping copies by the period rule) -- stream order, no queue, no barrier.
//   stored   stored blocks go from HBM to the window four bytes per lane, without the ring
//   output   every 8 KiB the new bytes are flushed to HBM as whole 4-byte words and summed into the stream's Adler-32
//   tables   lane 0 assigns the canonical codes of a dynamic / fixed block, all lanes fill the lookup tables
//
// The same source compiles for the host (NFI_HOST: one "lane", no barriers) so that tests/ can run it against zlib's own
// output on the CPU; on the device it is driven by nf_inflate.hip.  Every loop is bounded by the input and output
// lengths: malformed data ends with an error code, never with a wild access.
#pragma once
#include <stdint.h>

#ifdef NFI_HOST
#define NFI_UNROLL
#define NFI_NOUNROLL
#define NFI_FN static inline
#define NFI_CONST static const
#define NFI_LANE 0
#define NFI_NLANE 1
#define NFI_SYNC() ((void)0)
#else
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__GFX9__)
#error "nf_inflate_core.h: one stream = one 64-lane wavefront (gfx9 / CDNA); NFI_SYNC is no barrier on a wave32 target"
#endif
#define NFI_UNROLL _Pragma("unroll")
#define NFI_NOUNROLL _Pragma("nounroll")
#define NFI_FN __device__ inline
#define NFI_CONST __constant__ static const
#define NFI_LANE ((int)threadIdx.x)
#define NFI_NLANE 64
// One stream = one wavefront = one workgroup, so "all lanes have done their LDS writes" needs no s_barrier and no wait for
// the LDS queue to drain: a wavefront's LDS instructions execute in issue order, a later read sees an earlier write of
// another lane.  What is needed is that the compiler keeps that order: a wavefront-scope fence + scheduling barrier (no
// instruction is emitted).  k_inflate is launched with exactly 64 lanes per workgroup (nf_inflater_run checks warpSize).
#define NFI_SYNC()                                              \
    do {                                                        \
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  \
        __builtin_amdgcn_wave_barrier();                        \
    } while (0)
#endif
#define NFI_FOR_LANES(i, n) for (int i = NFI_LANE; i < (int)(n); i += NFI_NLANE)
// The symbol decoder is serial and every lane would compute the same thing, so it runs as UNIFORM code: all lanes execute it,
// every value it reads from LDS is passed through readfirstlane, and the compiler keeps the whole bit-buffer arithmetic on
// the scalar unit (native 64-bit shifts) instead of issuing 64-wide vector instructions for one useful lane.
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
    NFI_ERR_LAYOUT = 9,      // (device) the decoder state does not start at LDS offset 0
};

// RFC 1951 lets a match reach 32 KiB back: the LDS window holds the last kNfiWindow = 32 KiB of output, which makes the
// decoder state 39 KiB and FOUR streams per CU.  The window may be built smaller (a power of two >= 4 KiB): everything older
// than it has been flushed to the stream's output in HBM, and a match that reaches further back reads its bytes from there
// (nfi_copy_match: the far path).  Measured with 8 KiB (ten streams per CU, round 3, profiles/r03_inflate_window.txt): the
// per-stream rate HALVES -- zlib's matches in byte-shuffled model output reach back tens of grid rows, a far match costs
// 0.6-1.5 us (acquire + a round trip to L2 / HBM) against 0.3 us in LDS -- which eats the 2.5x in streams, and a group
// must hold 2560 chunks to fill the chip.  Not kept; the far path stays (and stays tested:
// tests/test_inflate_cpu.py runs the host build with -DNFI_WINDOW=8192 as well).
#ifndef NFI_WINDOW
#define NFI_WINDOW 32768
#endif
constexpr int kNfiWindow = NFI_WINDOW;
static_assert(kNfiWindow >= 4096 && kNfiWindow <= 32768 && (kNfiWindow & (kNfiWindow - 1)) == 0, "window: a power of two, 4 .. 32 KiB");
constexpr uint32_t kNfiMaxDist = 32768;
constexpr int kNfiRingWords = 256;         // input ring: two halves of 128 words (512 B each); a round ends when it runs low
constexpr int kNfiHalf = 128;
constexpr int kNfiLitBits = 10, kNfiDistBits = 8, kNfiClBits = 7;
constexpr uint32_t kNfiStoredRound = 4096; // bytes of a stored block moved per round (16 words per lane, all loads in flight)

// ---- lookup-table entries (32 bits): everything the decoder needs to know about the symbol at the head of the bit buffer
//   bits  0..3   n      bits of Huffman code this entry consumes (both codes of a literal pair); 0 with bit 31
//   bits  4..7   x      extra bits that follow the code (length codes 0..5, distance codes 0..13)
//   bits  8..23  v      literal (bits 8..15) and second literal (16..23) | length base 3..258 | distance base 1..24577
//   bits 24..25  lanes  literal entries: mask of the lanes that store (1 = one literal, 3 = two); 0 otherwise
//   bit  26      bad    a length / distance code the format reserves (286, 287 / 30, 31): an error once met
//   bit  28      a length code          | anything but a literal makes the entry >= 1 << 28:
//   bit  29      the end-of-block code  | that is the literal loop's whole exit test
//   bit  30      never in a table: the decoder ORs it into every entry it reads once the round has to end
//   bit  31      the code is longer than the table (literal / length table): canonical walk
constexpr uint32_t kNfiBad = 1u << 26, kNfiIsLen = 1u << 28, kNfiIsEob = 1u << 29, kNfiStop = 1u << 30, kNfiLong = 1u << 31;
constexpr uint32_t kNfiNotLit = 1u << 28;   // entries below this are literals
#define NFI_ENTRY(n, x, v, flags) ((uint32_t)(n) | ((uint32_t)(x) << 4) | ((uint32_t)(v) << 8) | (uint32_t)(flags))

template <int N> struct NfiHuffT {   // canonical code of one alphabet
    uint16_t count[16];              // number of codes of each length
    uint16_t offs[16];               // index in symbol[] of the first code of each length
    uint16_t first[16];              // canonical code (MSB first) of the first symbol of each length
    uint16_t symbol[N];              // symbols ordered by code
};
typedef NfiHuffT<288> NfiHuff;       // literal / length alphabet
typedef NfiHuffT<32> NfiHuffSmall;   // distance alphabet (30) and the code-length alphabet (19)

struct NfiCtx {           // lives in LDS (< 40 KiB, so that four fit a CU's 160 KiB): one per wavefront
    uint8_t window[kNfiWindow];
    uint32_t ring[kNfiRingWords];
    uint32_t lit_tab[1 << kNfiLitBits];
    uint32_t dist_tab[1 << kNfiDistBits];  // during a dynamic block's header its first bytes hold the code-length table
    NfiHuff lit;
    NfiHuffSmall dist;
    uint8_t lens[320];                     // code lengths: 288 literal/length + 32 distance
    uint32_t trash;                        // where the second lane of the literal store writes when there is one literal
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

// base and extra bits of the length codes 257..285 and the distance codes 0..29 (RFC 1951 3.2.5), by arithmetic
NFI_FN uint32_t nfi_len_entry(int li, int n)       // li = symbol - 257
{
    if (li >= 29) return NFI_ENTRY(n, 0, 3, kNfiIsLen | kNfiBad);
    const int lx = li < 8 || li == 28 ? 0 : (li - 4) >> 2;
    const uint32_t base = li < 8 ? 3u + (uint32_t)li : (li == 28 ? 258u : 3u + ((4u + (uint32_t)(li & 3)) << lx));
    return NFI_ENTRY(n, lx, base, kNfiIsLen);
}
NFI_FN uint32_t nfi_dist_entry(int ds, int n)
{
    if (ds >= 30) return NFI_ENTRY(n, 0, 1, kNfiBad);
    const int dx = ds < 4 ? 0 : (ds - 2) >> 1;
    const uint32_t base = ds < 4 ? 1u + (uint32_t)ds : 1u + ((2u + (uint32_t)(ds & 1)) << dx);
    return NFI_ENTRY(n, dx, base, 0);
}
NFI_FN uint32_t nfi_lit_entry(int sym, int n)
{
    if (sym < 256) return NFI_ENTRY(n, 0, sym, 1u << 24);
    if (sym == 256) return NFI_ENTRY(n, 0, 0, kNfiIsEob);
    return nfi_len_entry(sym - 257, n);
}

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

// canonical walk for a code that does not fit its lookup table (RFC 1951 3.2.2); consumes the code, returns the symbol
template <class H> NFI_FN int nfi_walk(NfiBits &b, const H &h)
{
    int code = 0, first = 0, index = 0;
    uint64_t bits = b.buf;
    NFI_NOUNROLL                                 // rare path: keep it small, and its table reads out of the hot loop
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
// lens[0..n) -> canonical description (one lane); returns "left" of the Kraft sum (0 = complete, > 0 = incomplete,
// < 0 = over-subscribed)
template <class H, class L> NFI_FN int nfi_canonical(H &h, const L *lens, int n)
{
    for (int l = 0; l <= 15; ++l) h.count[l] = 0;
    for (int s = 0; s < n; ++s) h.count[lens[s]]++;
    int left = 1;
    for (int l = 1; l <= 15; ++l) {
        left <<= 1;
        left -= h.count[l];
        if (left < 0) return left;
    }
    uint16_t offs[16];
    offs[0] = offs[1] = 0;
    h.offs[0] = h.offs[1] = 0;
    for (int l = 1; l < 15; ++l) {
        offs[l + 1] = (uint16_t)(offs[l] + h.count[l]);
        h.offs[l + 1] = offs[l + 1];
    }
    int cc = 0;                                // RFC 1951 3.2.2: code = (code + bl_count[bits-1]) << 1, bl_count[0] = 0
    h.first[0] = 0;
    for (int l = 1; l <= 15; ++l) {
        cc = (cc + (l > 1 ? h.count[l - 1] : 0)) << 1;
        h.first[l] = (uint16_t)cc;
    }
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l) h.symbol[offs[l]++] = (uint16_t)s;
    }
    return left;
}

NFI_FN uint32_t nfi_bitrev(uint32_t cw, int l)   // the stream carries Huffman codes most significant bit first
{
#ifdef NFI_HOST
    uint32_t r = 0;
    for (int k = 0; k < l; ++k) {
        r = (r << 1) | (cw & 1u);
        cw >>= 1;
    }
    return r;
#else
    return __builtin_bitreverse32(cw) >> (32 - l);
#endif
}

// all lanes: lookup table of `bits` bits for the alphabet described by h; LIT: literal/length entries, else distance entries.
// Lanes walk the symbols in code order: the i-th code of length l is first[l] + (i - offs[l]).
template <bool LIT, class H> NFI_FN void nfi_fill_table(uint32_t *tab, int bits, const H &h, int nused)
{
    NFI_FOR_LANES(k, 1 << bits) tab[k] = LIT ? kNfiLong : 0u;
    NFI_SYNC();
    NFI_FOR_LANES(i, nused) {
        int l = 1;
        while (l < 15 && i >= (int)h.offs[l] + (int)h.count[l]) ++l;
        if (l <= bits) {
            const int s = h.symbol[i];
            const uint32_t e = LIT ? nfi_lit_entry(s, l) : nfi_dist_entry(s, l);
            const uint32_t cw = nfi_bitrev((uint32_t)h.first[l] + (uint32_t)(i - (int)h.offs[l]), l);
            for (uint32_t k = cw; k < (1u << bits); k += (1u << l)) tab[k] = e;
        }
    }
    NFI_SYNC();
}

// all lanes: where two literals fit the index, the entry carries both (one lookup, two bytes).  The second literal of index
// k is what the single-symbol table says about the bits that follow the first code, provided its code ends inside the index.
NFI_FN void nfi_pair_literals(uint32_t *tab, int bits)
{
    constexpr int kPer = (1 << kNfiLitBits) / NFI_NLANE;
    uint32_t fresh[kPer];
NFI_UNROLL
    for (int r = 0; r < kPer; ++r) {
        const uint32_t k = (uint32_t)NFI_LANE + (uint32_t)r * NFI_NLANE;
        uint32_t e = tab[k];
        const uint32_t n1 = e & 15u;
        if (e < kNfiNotLit && (int)n1 < bits) {
            const uint32_t e2 = tab[k >> n1];
            const uint32_t n2 = e2 & 15u;
            if (e2 < kNfiNotLit && (int)(n1 + n2) <= bits)
                e = NFI_ENTRY(n1 + n2, 0, ((e >> 8) & 255u) | (((e2 >> 8) & 255u) << 8), 3u << 24);
        }
        fresh[r] = e;
    }
    NFI_SYNC();
NFI_UNROLL
    for (int r = 0; r < kPer; ++r) tab[(uint32_t)NFI_LANE + (uint32_t)r * NFI_NLANE] = fresh[r];
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
        for (int s = 0; s < 32; ++s) c.lens[288 + s] = 5;   // all 32 five-bit codes exist; 30 and 31 never occur in valid data
        c.nlit = 288;
        c.ndist = 32;
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
            uint16_t *cl_tab = reinterpret_cast<uint16_t *>(c.dist_tab);       // (symbol << 4) | length, 0 = none
            if (nfi_canonical(h, cl, 19) != 0 && !(h.count[0] == 18)) c.err = NFI_ERR_CODES;   // must be complete
            for (int k = 0; k < (1 << kNfiClBits); ++k) cl_tab[k] = 0;
            for (int i = 0; i < 19 - (int)h.count[0]; ++i) {
                int l = 1;
                while (l < 15 && i >= (int)h.offs[l] + (int)h.count[l]) ++l;
                const uint32_t cw = nfi_bitrev((uint32_t)h.first[l] + (uint32_t)(i - (int)h.offs[l]), l);
                for (uint32_t k = cw; k < (1u << kNfiClBits); k += (1u << l)) cl_tab[k] = (uint16_t)((h.symbol[i] << 4) | l);
            }
            int idx = 0;
            while (idx < nlit + ndist && !c.err) {
                nfi_refill(c, b);
                const uint32_t e = cl_tab[b.buf & ((1u << kNfiClBits) - 1u)];     // code lengths of this alphabet are <= 7 bits
                if (!(e & 15)) { c.err = NFI_ERR_SYMBOL; break; }
                nfi_take(b, (int)(e & 15));
                const int sym = (int)(e >> 4);
                if (sym < 16) {
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
        const int l1 = nfi_canonical(c.lit, c.lens, c.nlit);
        const int l2 = nfi_canonical(c.dist, c.lens + 288, c.ndist);
        // over-subscribed sets are errors; an incomplete literal/length set too, unless it has a single code; the
        // distance set may be incomplete (one distance code, or none when the block holds literals only)
        if (l1 < 0 || l2 < 0 || (l1 > 0 && c.nlit - c.lit.count[0] != 1)) c.err = NFI_ERR_CODES;
    }
    NFI_SYNC();
    if (c.err) return;                     // uniform
    nfi_fill_table<true>(c.lit_tab, kNfiLitBits, c.lit, c.nlit - (int)c.lit.count[0]);
    nfi_pair_literals(c.lit_tab, kNfiLitBits);
    nfi_fill_table<false>(c.dist_tab, kNfiDistBits, c.dist, c.ndist - (int)c.dist.count[0]);
}

// ---------------------------------------------------------------------------------------------- phase B: the symbol decoder
// A match is copied 64 bytes per step by all lanes; an overlapping copy (dist < len) repeats the last dist bytes.  A
// wavefront's LDS instructions execute in issue order, so a step may read what the previous step wrote.
NFI_FN void nfi_copy_match(NfiCtx &c, const uint8_t *flushed_out, uint32_t pos, uint32_t len, uint32_t dist)
{
    if (dist > (uint32_t)kNfiWindow) {
        // far: the source is older than the LDS window, i.e. it was flushed to the stream's output in HBM (a round never
        // leaves more than kNfiWindow - 258 bytes unflushed, so [pos - dist, pos - dist + len) lies below c.flushed); source
        // and destination cannot overlap.  The flush released its stores at agent scope; this acquires them.
#ifndef NFI_HOST
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
        NFI_FOR_LANES(j, len)
            c.window[(pos + (uint32_t)j) & (kNfiWindow - 1)] = flushed_out[pos + (uint32_t)j - dist];
        return;
    }
#ifndef NFI_HOST
    if (len <= (uint32_t)NFI_NLANE && dist >= len) {        // uniform; most matches of real data: one step, no loop
        if ((uint32_t)NFI_LANE < len)
            c.window[(pos + (uint32_t)NFI_LANE) & (kNfiWindow - 1)] = c.window[(pos + (uint32_t)NFI_LANE - dist) & (kNfiWindow - 1)];
        return;
    }
#endif
    if (dist >= len || dist >= (uint32_t)NFI_NLANE) {       // uniform: no lane reads a byte of its own step
        NFI_FOR_LANES(j, len)
            c.window[(pos + (uint32_t)j) & (kNfiWindow - 1)] = c.window[(pos + (uint32_t)j - dist) & (kNfiWindow - 1)];
        return;
    }
#ifdef NFI_HOST
    const float rcp = 1.0f / (float)dist;
#else
    const float rcp = __builtin_amdgcn_rcpf((float)dist);
#endif
    NFI_FOR_LANES(j, len) {                          // j < 258, dist < 64: the float quotient is exact to within one
        uint32_t off = (uint32_t)j;                  // the bytes repeat with period dist
        if (off >= dist) {
            const uint32_t q = (uint32_t)((float)off * rcp);
            off -= q * dist;
            if ((int32_t)off < 0) off += dist;
            else if (off >= dist) off -= dist;
        }
        c.window[(pos + (uint32_t)j) & (kNfiWindow - 1)] = c.window[(pos - dist + off) & (kNfiWindow - 1)];
    }
}

// Decode symbols of the current Huffman block straight into the LDS window until the block ends, the input ring runs low,
// or the window holds as much unflushed output as it can.  Executed by ALL lanes on uniform values (scalar code).
//
// A lone wavefront issues one instruction every four cycles, so after the table lookups what a symbol costs is its
// INSTRUCTION COUNT.  Hence:
//   * `e` always holds the entry of the symbol at the head of the bit buffer, requested one symbol ahead: the lookup is
//     issued, THEN the current symbol takes effect (a literal's byte store, a match's 64-lane copy), and only then is the
//     entry waited for -- the effect runs under the lookup's latency.  The same holds for the next word of the input ring.
//   * The run of table-coded literals (most symbols of real data) is a loop of its own with ONE exit test: literal entries
//     are the numbers below 1 << 28, and whatever ends a round (ring low, window full) ORs a stop bit into every entry read.
//   * Nothing in the hot path branches on an error: reserved codes carry a flag that is OR-ed up, a distance beyond the
//     start and an output beyond its length are noticed by sticky compares / at the end of the round.  Every index into
//     LDS is masked, so garbage decodes garbage safely until the round ends (each symbol consumes at least one bit).
NFI_FN void nfi_decode_round_cxx(NfiCtx &c, const uint8_t *flushed_out, uint32_t out_len)
{
    NfiBits b;
    b.buf = ((uint64_t)NFI_UNI((uint32_t)(c.bitbuf >> 32)) << 32) | NFI_UNI((uint32_t)c.bitbuf);
    b.cnt = (int)NFI_UNI(c.bitcnt);
    b.word = NFI_UNI(c.word);
    const int last = (int)NFI_UNI(c.last);
    // ring words [b.word, loaded) are valid; one symbol reads at most three of them, the prefetch one more
    const uint32_t word_stop = NFI_UNI(c.loaded) - 5u;
    // The unflushed output must stay inside the window.  The position is looked at after every match and at every refill of
    // the bit buffer, i.e. at least every 32 bits = 32 bytes of literals (one-bit codes): a round ends at least one match
    // (258) and one such run short of the window.
    const uint32_t hard_stop = NFI_UNI(c.flushed) + (uint32_t)kNfiWindow - 512u;
    uint32_t nextw_raw = c.ring[b.word & (kNfiRingWords - 1)];          // not waited for until the next refill
    uint32_t stop = 0;                            // kNfiStop once the round has to end: OR-ed into every entry read
#define NFI_REFILL()                                                     \
    if (b.cnt <= 32) {                                                   \
        b.buf |= (uint64_t)NFI_UNI(nextw_raw) << b.cnt;                  \
        b.cnt += 32;                                                     \
        ++b.word;                                                        \
        nextw_raw = c.ring[b.word & (kNfiRingWords - 1)];                \
        if (b.word >= word_stop || pos >= hard_stop) stop = kNfiStop;    \
    }
    constexpr uint32_t kMask = (1u << kNfiLitBits) - 1u;
#define NFI_LOOKUP_RAW() (c.lit_tab[(uint32_t)b.buf & kMask])
    uint32_t pos = NFI_UNI(c.pos);
    uint32_t flags = 0;                           // OR of every length / distance entry used: kNfiBad = a reserved code
    int err = 0, state = 1;
    if (pos >= hard_stop) stop = kNfiStop;
    NFI_REFILL();
    uint32_t e = NFI_UNI(NFI_LOOKUP_RAW()) | stop;
    for (;;) {
        // ---- run of table-coded literals (one or two per lookup)
        while (e < kNfiNotLit) {
            const uint32_t n = e & 15u;
            b.buf >>= n;
            b.cnt -= (int)n;
            NFI_REFILL();
            const uint32_t raw_next = NFI_LOOKUP_RAW();
            const uint32_t count = (((e >> 24) & 3u) + 1u) >> 1;     // lane mask 1 / 3 -> 1 / 2 literals
#ifdef NFI_HOST
            c.window[pos & (kNfiWindow - 1)] = (uint8_t)(e >> 8);
            if (count > 1u) c.window[(pos + 1u) & (kNfiWindow - 1)] = (uint8_t)(e >> 16);
#else
            if (NFI_LANE < 2) {                   // one store instruction for both bytes; lane 1 of a single literal hits `trash`
                uint8_t *at = (uint32_t)NFI_LANE < count ? &c.window[(pos + (uint32_t)NFI_LANE) & (kNfiWindow - 1)]
                                                         : reinterpret_cast<uint8_t *>(&c.trash);
                *at = (uint8_t)(e >> (8u + 8u * (uint32_t)NFI_LANE));
            }
#endif
            pos += count;
            e = NFI_UNI(raw_next) | stop;
        }
        if (stop) break;                          // the caller refills the ring / flushes, then comes back
        // ---- anything else: a match, the end of the block, a code longer than the table
        uint32_t n = e & 15u;
        if (e >= kNfiIsEob) {                     // rare: end of block or long code
            if (e & kNfiIsEob) {
                b.buf >>= n;
                b.cnt -= (int)n;
                state = last ? 3 : 0;
                break;
            }
            const int sym = nfi_walk(b, c.lit);   // consumes the code
            if (sym < 0) { err = NFI_ERR_SYMBOL; break; }
            e = nfi_lit_entry(sym, 0);
            n = 0;
            if (e < kNfiNotLit) {                 // a literal with a long code
                NFI_REFILL();
                const uint32_t raw_next = NFI_LOOKUP_RAW();
                c.window[pos & (kNfiWindow - 1)] = (uint8_t)(e >> 8);
                pos += 1u;
                e = NFI_UNI(raw_next) | stop;
                continue;
            }
            if (e & kNfiIsEob) {
                state = last ? 3 : 0;
                break;
            }
        }
        // ---- a match: length (base + extra bits), then the distance code
        b.buf >>= n;
        b.cnt -= (int)n;
        flags |= e;
        const uint32_t len = ((e >> 8) & 0xffffu) + nfi_take(b, (int)((e >> 4) & 15u));
        NFI_REFILL();
        uint32_t d = NFI_UNI(c.dist_tab[(uint32_t)b.buf & ((1u << kNfiDistBits) - 1u)]);
        uint32_t dn = d & 15u;
        if (dn == 0) {
            const int ds = nfi_walk(b, c.dist);
            if (ds < 0) { err = NFI_ERR_SYMBOL; break; }
            d = nfi_dist_entry(ds, 0);
        }
        b.buf >>= dn;
        b.cnt -= (int)dn;
        flags |= d;
        const uint32_t dist = ((d >> 8) & 0xffffu) + nfi_take(b, (int)((d >> 4) & 15u));
        if (dist > pos) err = NFI_ERR_DISTANCE;     // sticky; the copy below reads masked (stale) window bytes
        if (dist > pos) break;                      // (a far copy would read before the start of the output)
        NFI_REFILL();
        const uint32_t raw_next = NFI_LOOKUP_RAW();
        nfi_copy_match(c, flushed_out, pos, len, dist);          // len in 3 .. 258, dist in 1 .. 32768
        pos += len;
        if (pos >= hard_stop) stop = kNfiStop;
        e = NFI_UNI(raw_next) | stop;
    }
#undef NFI_LOOKUP_RAW
#undef NFI_REFILL
    if (!err && (flags & kNfiBad)) err = NFI_ERR_SYMBOL;
    if (!err && pos > out_len) err = NFI_ERR_OUTPUT;
    if (err) c.err = err;
    c.state = state;
    c.pos = pos;
    c.bitbuf = b.buf;
    c.bitcnt = b.cnt;
    c.word = b.word;
}

#if !defined(NFI_HOST) && !defined(NFI_PORTABLE_DECODE)
// ---- the same round, with its hot loop written in gfx9 ISA ------------------------------------------------------------
// What the compiler makes of nfi_decode_round_cxx costs ~27 instructions and three taken branches per literal lookup and
// ~100 instructions and a dozen taken branches per match; for a lone wavefront every instruction is >= 4 cycles, every taken
// branch ~21 and every dependent LDS lookup ~116 (tools/lds_chain.hip).  Below:
//   * the LOOKUP TABLES LIVE IN REGISTERS inside the block: 1024 literal / length entries = v60..v75 x 64 lanes, 256 distance
//     entries = v76..v79 (loaded from the LDS tables on entry).  VGPR index mode is ON for the whole block with M0 = the row
//     of the pending lookup, so a lookup is  s_set_gpr_idx_idx (row)  +  v_readlane ..., v60 / v76, lane  -- no LDS round
//     trip in the chain.  Every other vector instruction of the block takes an SGPR or a constant as its first source (only
//     VGPR first sources are indexed); the refill, which reads one, sets the index to 0 first.
//   * a literal lookup is 19 instructions and ONE taken branch: both bytes of a pair go out in one ds_write_b8 (lanes 0 and 1,
//     exec_lo straight from the entry); literal entries are the numbers below s39, which drops to 0 when the round has to end;
//   * a short match is ~55 instructions and two taken branches (s_bfm_b64 exec = the copy's lanes); copies of 64 bytes and
//     more loop out of line; the only LDS wait in the block is a copy's own read-before-write;
//   * the block runs literal runs and matches with table-coded length and distance whose source lies before the destination
//     (and inside the LDS window); everything rare leaves the block at a symbol boundary with a reason code and is finished
//     by the C++ statements of the portable version:
//       reason 0  the round has to end (ring low / window full)        1  head entry is end-of-block or a long code
//              2  distance code longer than its table (length decoded)  3  overlapping or far copy (len, dist decoded)
// Register plan inside the block (moved in and out at its ends): s39 literal limit, s[40:41] bit buffer, s42 valid bits, s43
// entry at the head, s44 output position, s45 ring word, s46 stop bit, s47-s49 scratch, s50 len, s51 dist, s52 distance
// entry, s53 OR of the entries used, s54 error, s55 / s56 word / position limits, s57 reason, s[58:59] saved exec (s38: saved
// M0 -- index mode overwrites it, and a reserved register cannot go on the clobber list), s60 / s61
// row / lane of the pending lookup, s[62:63] lane mask of a literal store; v40-v48 scratch, v60-v79 the tables.
// ctx sits at LDS address 0 (k_inflate checks): the window is addressed from 0, the tables by immediate offsets.
NFI_FN void nfi_decode_round_asm(NfiCtx &c, const uint8_t *flushed_out, uint32_t out_len)
{
    uint64_t buf = ((uint64_t)NFI_UNI((uint32_t)(c.bitbuf >> 32)) << 32) | NFI_UNI((uint32_t)c.bitbuf);
    uint32_t cnt = NFI_UNI(c.bitcnt), word = NFI_UNI(c.word), pos = NFI_UNI(c.pos);
    const int last = (int)NFI_UNI(c.last);
    const uint32_t word_stop = NFI_UNI(c.loaded) - 5u;
    const uint32_t hard_stop = NFI_UNI(c.flushed) + (uint32_t)kNfiWindow - 512u;
    uint32_t nextw_raw = c.ring[word & (kNfiRingWords - 1)];
    uint32_t stop = pos >= hard_stop ? kNfiStop : 0u;
    uint32_t flags = 0, err = 0, reason = 0, len = 0, dist = 0;
    int state = 1;
    constexpr uint32_t kMask = (1u << kNfiLitBits) - 1u;
    const uint32_t lane = (uint32_t)NFI_LANE;
#define NFI_REFILL_C()                                                   \
    if (cnt <= 32u) {                                                    \
        buf |= (uint64_t)NFI_UNI(nextw_raw) << cnt;                      \
        cnt += 32u;                                                      \
        ++word;                                                          \
        nextw_raw = c.ring[word & (kNfiRingWords - 1)];                  \
        if (word >= word_stop || pos >= hard_stop) stop = kNfiStop;      \
    }
    NFI_REFILL_C();
    uint32_t e = NFI_UNI(c.lit_tab[(uint32_t)buf & kMask]) | stop;
    for (;;) {
        // (what enters the block through an "s" operand must be uniform for the compiler too, not just in fact)
        buf = ((uint64_t)NFI_UNI((uint32_t)(buf >> 32)) << 32) | NFI_UNI((uint32_t)buf);
        cnt = NFI_UNI(cnt); e = NFI_UNI(e); pos = NFI_UNI(pos); word = NFI_UNI(word); stop = NFI_UNI(stop);
        flags = NFI_UNI(flags); err = NFI_UNI(err);
        asm volatile(
            "s_mov_b64 s[58:59], exec\n\t"
            "s_mov_b32 s38, m0\n\t"            // VGPR index mode below overwrites M0: the compiler's value is put back at the end
            "s_mov_b64 s[40:41], %[buf]\n\t"
            "s_mov_b32 s42, %[cnt]\n\t"
            "s_mov_b32 s43, %[e]\n\t"
            "s_mov_b32 s44, %[pos]\n\t"
            "s_mov_b32 s45, %[word]\n\t"
            "s_mov_b32 s46, %[stop]\n\t"
            "s_mov_b32 s53, %[flags]\n\t"
            "s_mov_b32 s54, %[err]\n\t"
            "s_mov_b32 s55, %[word_stop]\n\t"
            "s_mov_b32 s56, %[hard_stop]\n\t"
            "s_mov_b32 s57, 0\n\t"
            "s_mov_b32 s50, 0\n\t"
            "s_mov_b32 s51, 0\n\t"
            "s_mov_b32 s63, 0\n\t"
            "s_cmp_eq_u32 s46, 0\n\t"
            "s_cselect_b32 s39, 0x10000000, 0\n\t"
            "s_mov_b64 exec, -1\n\t"
            "v_lshlrev_b32 v48, 3, %[lane]\n\t"
            "v_add_u32 v48, 8, v48\n\t"
            "v_lshlrev_b32 v40, 2, %[lane]\n\t"
            // The lookup tables move into registers: 1024 literal / length entries = v60..v75 x 64 lanes, 256 distance entries =
            // v76..v79.  VGPR index mode stays ON for the whole block with M0 = the row of the pending lookup, so a lookup is
            // s_set_gpr_idx_idx (row) + v_readlane ..., v60 / v76, lane: no LDS round trip in the chain (tools/lds_chain.hip: 33 ns
            // against 50).  While the mode is on every other vector instruction of the block takes an SGPR or a constant as its
            // first source (only VGPR first sources are indexed); the refill, which reads one, sets the index to 0 first.
            "ds_read_b32 v60, v40 offset:%[lit]+0\n\t"
            "ds_read_b32 v61, v40 offset:%[lit]+256\n\t"
            "ds_read_b32 v62, v40 offset:%[lit]+512\n\t"
            "ds_read_b32 v63, v40 offset:%[lit]+768\n\t"
            "ds_read_b32 v64, v40 offset:%[lit]+1024\n\t"
            "ds_read_b32 v65, v40 offset:%[lit]+1280\n\t"
            "ds_read_b32 v66, v40 offset:%[lit]+1536\n\t"
            "ds_read_b32 v67, v40 offset:%[lit]+1792\n\t"
            "ds_read_b32 v68, v40 offset:%[lit]+2048\n\t"
            "ds_read_b32 v69, v40 offset:%[lit]+2304\n\t"
            "ds_read_b32 v70, v40 offset:%[lit]+2560\n\t"
            "ds_read_b32 v71, v40 offset:%[lit]+2816\n\t"
            "ds_read_b32 v72, v40 offset:%[lit]+3072\n\t"
            "ds_read_b32 v73, v40 offset:%[lit]+3328\n\t"
            "ds_read_b32 v74, v40 offset:%[lit]+3584\n\t"
            "ds_read_b32 v75, v40 offset:%[lit]+3840\n\t"
            "ds_read_b32 v76, v40 offset:%[dtab]+0\n\t"
            "ds_read_b32 v77, v40 offset:%[dtab]+256\n\t"
            "ds_read_b32 v78, v40 offset:%[dtab]+512\n\t"
            "ds_read_b32 v79, v40 offset:%[dtab]+768\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b64 exec, 3\n\t"
            "s_set_gpr_idx_on s63, 0x1\n\t"
            // ---------------------------------------------------------------- dispatch on the entry at the head
            "1:\n"
            "s_cmp_lt_u32 s43, s39\n\t"
            "s_cbranch_scc0 3f\n\t"
            // ---------------------------------------------------------------- literal run: lanes 0 and 1 store, exec_hi stays 0
            "2:\n"
            "s_and_b32 s47, s43, 15\n\t"
            "s_lshr_b64 s[40:41], s[40:41], s47\n\t"
            "s_sub_u32 s42, s42, s47\n\t"
            "s_cmp_le_u32 s42, 32\n\t"
            "s_cbranch_scc1 10f\n\t"
            "11:\n"
            "s_bfe_u32 s60, s40, 0x40006\n\t"
            "s_and_b32 s61, s40, 63\n\t"
            "s_set_gpr_idx_idx s60\n\t"
            "s_bfe_u32 s62, s43, 0x20018\n\t"
            "s_mov_b32 exec_lo, s62\n\t"
            "v_add_u32 v42, s44, %[lane]\n\t"
            "v_and_b32 v42, %[wmask], v42\n\t"
            "v_bfe_u32 v44, s43, v48, 8\n\t"
            "ds_write_b8 v42, v44\n\t"
            "s_bcnt1_i32_b32 s47, s62\n\t"
            "s_add_u32 s44, s44, s47\n\t"
            "v_readlane_b32 s43, v60, s61\n\t"
            "s_cmp_lt_u32 s43, s39\n\t"
            "s_cbranch_scc1 2b\n\t"
            // ---------------------------------------------------------------- not a literal (or the round has to end)
            "3:\n"
            "s_cmp_lg_u32 s46, 0\n\t"
            "s_cbranch_scc1 9f\n\t"
            "s_cmp_ge_u32 s43, 0x20000000\n\t"
            "s_cbranch_scc1 20f\n\t"
            // ---- a match: length
            "s_and_b32 s47, s43, 15\n\t"
            "s_lshr_b64 s[40:41], s[40:41], s47\n\t"
            "s_sub_u32 s42, s42, s47\n\t"
            "s_or_b32 s53, s53, s43\n\t"
            "s_bfe_u32 s47, s43, 0x40004\n\t"
            "s_bfm_b32 s48, s47, 0\n\t"
            "s_and_b32 s48, s40, s48\n\t"
            "s_lshr_b64 s[40:41], s[40:41], s47\n\t"
            "s_sub_u32 s42, s42, s47\n\t"
            "s_bfe_u32 s50, s43, 0x100008\n\t"
            "s_add_u32 s50, s50, s48\n\t"
            "s_cmp_le_u32 s42, 32\n\t"
            "s_cbranch_scc1 12f\n\t"
            // ---- distance (table in v76..v79)
            "13:\n"
            "s_bfe_u32 s60, s40, 0x20006\n\t"
            "s_and_b32 s61, s40, 63\n\t"
            "s_set_gpr_idx_idx s60\n\t"
            "v_readlane_b32 s52, v76, s61\n\t"
            "s_and_b32 s47, s52, 15\n\t"
            "s_cmp_eq_u32 s47, 0\n\t"
            "s_cbranch_scc1 21f\n\t"
            "s_lshr_b64 s[40:41], s[40:41], s47\n\t"
            "s_sub_u32 s42, s42, s47\n\t"
            "s_or_b32 s53, s53, s52\n\t"
            "s_bfe_u32 s47, s52, 0x40004\n\t"
            "s_bfm_b32 s48, s47, 0\n\t"
            "s_and_b32 s48, s40, s48\n\t"
            "s_lshr_b64 s[40:41], s[40:41], s47\n\t"
            "s_sub_u32 s42, s42, s47\n\t"
            "s_bfe_u32 s51, s52, 0x100008\n\t"
            "s_add_u32 s51, s51, s48\n\t"
            "s_cmp_gt_u32 s51, s44\n\t"
            "s_cselect_b32 s54, 5, s54\n\t"
            "s_cmp_le_u32 s42, 32\n\t"
            "s_cbranch_scc1 14f\n\t"
            "15:\n"
            "s_cmp_lt_u32 s51, s50\n\t"
            "s_cbranch_scc1 22f\n\t"
            ".if %[win] < 32768\n"
            "s_cmp_gt_u32 s51, %[win]\n\t"
            "s_cbranch_scc1 22f\n\t"
            ".endif\n"
            "s_cmp_ge_u32 s50, 64\n\t"
            "s_cbranch_scc1 17f\n\t"
            // ---- the copy, one step of len < 64 lanes (source before destination: dist >= len here), then the next entry
            "s_bfm_b64 exec, s50, 0\n\t"
            "v_add_u32 v42, s44, %[lane]\n\t"
            "v_subrev_u32 v43, s51, v42\n\t"
            "v_and_b32 v43, %[wmask], v43\n\t"
            "v_and_b32 v42, %[wmask], v42\n\t"
            "ds_read_u8 v44, v43\n\t"
            "s_add_u32 s44, s44, s50\n\t"
            "s_cmp_ge_u32 s44, s56\n\t"
            "s_cselect_b32 s46, 0x40000000, s46\n\t"
            "s_cselect_b32 s39, 0, s39\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_write_b8 v42, v44\n\t"
            "s_mov_b64 exec, 3\n\t"
            "s_bfe_u32 s60, s40, 0x40006\n\t"
            "s_and_b32 s61, s40, 63\n\t"
            "s_set_gpr_idx_idx s60\n\t"
            "v_readlane_b32 s43, v60, s61\n\t"
            "s_cmp_lt_u32 s43, s39\n\t"
            "s_cbranch_scc1 2b\n\t"
            "s_branch 3b\n\t"
            // ---- copies of 64 bytes and more: steps of 64 lanes (out of line)
            "17:\n"
            "s_mov_b64 exec, -1\n\t"
            "v_add_u32 v42, s44, %[lane]\n\t"
            "v_subrev_u32 v43, s51, v42\n\t"
            "s_add_u32 s44, s44, s50\n\t"
            "s_cmp_ge_u32 s44, s56\n\t"
            "s_cselect_b32 s46, 0x40000000, s46\n\t"
            "s_cselect_b32 s39, 0, s39\n\t"
            "18:\n"
            "s_min_u32 s47, s50, 64\n\t"
            "s_bfm_b64 exec, s47, 0\n\t"
            "s_cmp_ge_u32 s50, 64\n\t"
            "s_cselect_b64 exec, -1, exec\n\t"
            "v_and_b32 v46, %[wmask], v43\n\t"
            "v_and_b32 v45, %[wmask], v42\n\t"
            "ds_read_u8 v44, v46\n\t"
            "s_sub_u32 s50, s50, s47\n\t"
            "v_add_u32 v42, 64, v42\n\t"
            "v_add_u32 v43, 64, v43\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "ds_write_b8 v45, v44\n\t"
            "s_cmp_lg_u32 s50, 0\n\t"
            "s_cbranch_scc1 18b\n\t"
            "s_mov_b64 exec, 3\n\t"
            "s_bfe_u32 s60, s40, 0x40006\n\t"
            "s_and_b32 s61, s40, 63\n\t"
            "s_set_gpr_idx_idx s60\n\t"
            "v_readlane_b32 s43, v60, s61\n\t"
            "s_cmp_lt_u32 s43, s39\n\t"
            "s_cbranch_scc1 2b\n\t"
            "s_branch 3b\n\t"
            // ---------------------------------------------------------------- refills of the bit buffer: out of line, one copy per
            // site (a taken branch costs a lone wavefront ~21 cycles -- tools/lds_chain.hip -- so no shared trampoline)
            "10:\n"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_set_gpr_idx_idx s63\n\t"
            "v_readfirstlane_b32 s48, %[nextw]\n\t"
            "s_mov_b32 s49, 0\n\t"
            "s_lshl_b64 s[48:49], s[48:49], s42\n\t"
            "s_or_b64 s[40:41], s[40:41], s[48:49]\n\t"
            "s_add_u32 s42, s42, 32\n\t"
            "s_add_u32 s45, s45, 1\n\t"
            "s_and_b32 s47, s45, 0xff\n\t"
            "s_lshl_b32 s47, s47, 2\n\t"
            "v_mov_b32 v40, s47\n\t"
            "ds_read_b32 %[nextw], v40 offset:%[ring]\n\t"
            "s_cmp_ge_u32 s45, s55\n\t"
            "s_cselect_b32 s46, 0x40000000, s46\n\t"
            "s_cselect_b32 s39, 0, s39\n\t"
            "s_cmp_ge_u32 s44, s56\n\t"
            "s_cselect_b32 s46, 0x40000000, s46\n\t"
            "s_cselect_b32 s39, 0, s39\n\t"
            "s_branch 11b\n\t"
            "12:\n"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_set_gpr_idx_idx s63\n\t"
            "v_readfirstlane_b32 s48, %[nextw]\n\t"
            "s_mov_b32 s49, 0\n\t"
            "s_lshl_b64 s[48:49], s[48:49], s42\n\t"
            "s_or_b64 s[40:41], s[40:41], s[48:49]\n\t"
            "s_add_u32 s42, s42, 32\n\t"
            "s_add_u32 s45, s45, 1\n\t"
            "s_and_b32 s47, s45, 0xff\n\t"
            "s_lshl_b32 s47, s47, 2\n\t"
            "v_mov_b32 v40, s47\n\t"
            "ds_read_b32 %[nextw], v40 offset:%[ring]\n\t"
            "s_cmp_ge_u32 s45, s55\n\t"
            "s_cselect_b32 s46, 0x40000000, s46\n\t"
            "s_cselect_b32 s39, 0, s39\n\t"
            "s_cmp_ge_u32 s44, s56\n\t"
            "s_cselect_b32 s46, 0x40000000, s46\n\t"
            "s_cselect_b32 s39, 0, s39\n\t"
            "s_branch 13b\n\t"
            "14:\n"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_set_gpr_idx_idx s63\n\t"
            "v_readfirstlane_b32 s48, %[nextw]\n\t"
            "s_mov_b32 s49, 0\n\t"
            "s_lshl_b64 s[48:49], s[48:49], s42\n\t"
            "s_or_b64 s[40:41], s[40:41], s[48:49]\n\t"
            "s_add_u32 s42, s42, 32\n\t"
            "s_add_u32 s45, s45, 1\n\t"
            "s_and_b32 s47, s45, 0xff\n\t"
            "s_lshl_b32 s47, s47, 2\n\t"
            "v_mov_b32 v40, s47\n\t"
            "ds_read_b32 %[nextw], v40 offset:%[ring]\n\t"
            "s_cmp_ge_u32 s45, s55\n\t"
            "s_cselect_b32 s46, 0x40000000, s46\n\t"
            "s_cselect_b32 s39, 0, s39\n\t"
            "s_cmp_ge_u32 s44, s56\n\t"
            "s_cselect_b32 s46, 0x40000000, s46\n\t"
            "s_cselect_b32 s39, 0, s39\n\t"
            "s_branch 15b\n\t"
            // ---------------------------------------------------------------- exits
            "20:\n"
            "s_mov_b32 s57, 1\n\t"
            "s_branch 8f\n\t"
            "21:\n"
            "s_mov_b32 s57, 2\n\t"
            "s_branch 8f\n\t"
            "22:\n"
            "s_mov_b32 s57, 3\n\t"
            "s_branch 8f\n\t"
            "9:\n"
            "s_mov_b32 s57, 0\n\t"
            "8:\n"
            "s_set_gpr_idx_off\n\t"
            "s_mov_b32 m0, s38\n\t"
            "s_waitcnt lgkmcnt(0)\n\t"
            "s_mov_b64 exec, s[58:59]\n\t"
            "s_mov_b64 %[buf], s[40:41]\n\t"
            "s_mov_b32 %[cnt], s42\n\t"
            "s_or_b32 s43, s43, s46\n\t"
            "s_mov_b32 %[e], s43\n\t"
            "s_mov_b32 %[pos], s44\n\t"
            "s_mov_b32 %[word], s45\n\t"
            "s_mov_b32 %[stop], s46\n\t"
            "s_mov_b32 %[flags], s53\n\t"
            "s_mov_b32 %[err], s54\n\t"
            "s_mov_b32 %[reason], s57\n\t"
            "s_mov_b32 %[len], s50\n\t"
            "s_mov_b32 %[dist], s51\n\t"
            : [buf] "+s"(buf), [cnt] "+s"(cnt), [e] "+s"(e), [pos] "+s"(pos), [word] "+s"(word), [stop] "+s"(stop),
              [flags] "+s"(flags), [err] "+s"(err), [reason] "=s"(reason), [len] "=s"(len), [dist] "=s"(dist),
              [nextw] "+v"(nextw_raw)
            : [word_stop] "s"(word_stop), [hard_stop] "s"(hard_stop), [lane] "v"(lane),
              [lit] "n"(__builtin_offsetof(NfiCtx, lit_tab)), [dtab] "n"(__builtin_offsetof(NfiCtx, dist_tab)),
              [ring] "n"(__builtin_offsetof(NfiCtx, ring)), [win] "n"(kNfiWindow), [wmask] "n"(kNfiWindow - 1)
            : "memory", "scc", "vcc", "s38", "s39", "s40", "s41", "s42", "s43", "s44", "s45", "s46", "s47", "s48", "s49", "s50", "s51",
              "s52", "s53", "s54", "s55", "s56", "s57", "s58", "s59", "s60", "s61", "s62", "s63", "v40", "v41", "v42", "v43", "v44",
              "v45", "v46", "v48", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73",
              "v74", "v75", "v76", "v77", "v78", "v79");
        // ---- the block left at a symbol boundary: the rare cases, in the portable version's statements
        if (reason == 0) break;
        if (reason == 1) {                           // end of block, or a literal / length code longer than the table
            uint32_t n = e & 15u;
            NfiBits b{buf, (int)cnt, word};
            if (e & kNfiIsEob) {
                buf >>= n;
                cnt -= n;
                state = last ? 3 : 0;
                break;
            }
            const int sym = nfi_walk(b, c.lit);
            buf = b.buf;
            cnt = (uint32_t)b.cnt;
            if (sym < 0) { err = NFI_ERR_SYMBOL; break; }
            e = nfi_lit_entry(sym, 0);
            if (e < kNfiNotLit) {
                NFI_REFILL_C();
                const uint32_t raw_next = c.lit_tab[(uint32_t)buf & kMask];
                c.window[pos & (kNfiWindow - 1)] = (uint8_t)(e >> 8);
                pos += 1u;
                e = NFI_UNI(raw_next) | stop;
                continue;
            }
            if (e & kNfiIsEob) {
                state = last ? 3 : 0;
                break;
            }
            flags |= e;                              // a length: finish the match here
            const uint32_t x = (e >> 4) & 15u;
            len = ((e >> 8) & 0xffffu) + ((uint32_t)buf & ((1u << x) - 1u));
            buf >>= x;
            cnt -= x;
            NFI_REFILL_C();
            reason = 2;
        }
        if (reason == 2) {                           // the distance: table or canonical walk
            uint32_t d = NFI_UNI(c.dist_tab[(uint32_t)buf & ((1u << kNfiDistBits) - 1u)]);
            const uint32_t dn = d & 15u;
            if (dn == 0) {
                NfiBits b{buf, (int)cnt, word};
                const int ds = nfi_walk(b, c.dist);
                buf = b.buf;
                cnt = (uint32_t)b.cnt;
                if (ds < 0) { err = NFI_ERR_SYMBOL; break; }
                d = nfi_dist_entry(ds, 0);
            }
            buf >>= dn;
            cnt -= dn;
            flags |= d;
            const uint32_t dx = (d >> 4) & 15u;
            dist = ((d >> 8) & 0xffffu) + ((uint32_t)buf & ((1u << dx) - 1u));
            buf >>= dx;
            cnt -= dx;
            if (dist > pos) { err = NFI_ERR_DISTANCE; break; }
            NFI_REFILL_C();
        }
        if (err) break;                              // (reason 3 with the sticky distance error: no far read before the start)
        // reason 2 (continued) and 3: the copy in its general form, then the next entry
        const uint32_t raw_next = c.lit_tab[(uint32_t)buf & kMask];
        nfi_copy_match(c, flushed_out, pos, len, dist);
        pos += len;
        if (pos >= hard_stop) stop = kNfiStop;
        e = NFI_UNI(raw_next) | stop;
    }
#undef NFI_REFILL_C
    if (!err && (flags & kNfiBad)) err = NFI_ERR_SYMBOL;
    if (!err && pos > out_len) err = NFI_ERR_OUTPUT;
    if (err) c.err = (int32_t)err;
    c.state = state;
    c.pos = pos;
    c.bitbuf = buf;
    c.bitcnt = (int32_t)cnt;
    c.word = word;
}
#define nfi_decode_round nfi_decode_round_asm
#else
#define nfi_decode_round nfi_decode_round_cxx
#endif

// ---------------------------------------------------------------------------------------------- phase C: all lanes
// bytes of a stored block: straight from HBM into the window, four bytes per lane and step (the header left the bit buffer
// on a byte boundary; whatever it still holds is read again from memory).  When the block ends the bit buffer restarts at
// the byte that follows it and the input ring is re-positioned there.
NFI_FN void nfi_stored_round(NfiCtx &c, const uint32_t *words, uint32_t nwords, uint32_t out_len)
{
    const uint32_t n = c.stored_left < kNfiStoredRound ? c.stored_left : kNfiStoredRound;
    const uint32_t byte_in = c.word * 4u - ((uint32_t)c.bitcnt >> 3);     // next unread byte of the stream
    const uint32_t pos = c.pos;
    int err = 0;
    if (pos + n > out_len) err = NFI_ERR_OUTPUT;
    else if (byte_in + n > nwords * 4u) err = NFI_ERR_INPUT;
    if (err) {                                                            // uniform
        if (NFI_LANE == 0) c.err = err;
        NFI_SYNC();
        return;
    }
    const uint32_t sh = 8u * (byte_in & 3u), w0 = byte_in >> 2, nw = (n + 3u) >> 2;
    constexpr int kPer = (int)(kNfiStoredRound / 4u) / NFI_NLANE;        // words per lane and round: all loads issued first
    uint32_t lo[kPer], hi[kPer];
    NFI_UNROLL
    for (int r = 0; r < kPer; ++r) {
        const uint32_t k = (uint32_t)NFI_LANE + (uint32_t)r * NFI_NLANE;
        lo[r] = k < nw ? words[w0 + k] : 0u;
        hi[r] = (sh && k < nw && w0 + k + 1u < nwords) ? words[w0 + k + 1u] : 0u;
    }
    NFI_UNROLL
    for (int r = 0; r < kPer; ++r) {
        const uint32_t k = (uint32_t)NFI_LANE + (uint32_t)r * NFI_NLANE;
        if (k >= nw) break;
        const uint32_t v = sh ? (lo[r] >> sh) | (hi[r] << (32u - sh)) : lo[r];
        const uint32_t p = pos + 4u * k;
        if ((pos & 3u) == 0 && 4u * k + 4u <= n) {
            *reinterpret_cast<uint32_t *>(&c.window[p & (kNfiWindow - 1)]) = v;
        } else {
            for (uint32_t q = 0; q < 4; ++q)
                if (4u * k + q < n) c.window[(p + q) & (kNfiWindow - 1)] = (uint8_t)(v >> (8u * q));
        }
    }
    NFI_SYNC();
    if (NFI_LANE == 0) {
        c.pos = pos + n;
        c.stored_left -= n;
        const uint32_t bi = byte_in + n;                                  // first unread byte
        uint32_t w = bi >> 2;
        const uint32_t sub = bi & 3u;
        if (sub) {                                                        // keep the rest of that word in the bit buffer
            c.bitbuf = (uint64_t)((w < nwords ? words[w] : 0u) >> (8u * sub));
            c.bitcnt = 32 - 8 * (int)sub;
            ++w;
        } else {
            c.bitbuf = 0;
            c.bitcnt = 0;
        }
        c.word = w;
        c.loaded = w & ~(uint32_t)(kNfiHalf - 1);                         // the ring is filled again from the half that holds w
        if (c.stored_left == 0) c.state = c.last ? 3 : 0;
    }
    NFI_SYNC();
}

// write the finished part of the window to the output: whole 4-byte words (dst 4-byte aligned), the tail at the end;
// the Adler-32 of the stream (RFC 1950) is carried along: for n new bytes b_0..b_{n-1}, a += sum b_i and
// b += n*a_old + sum (n-i) b_i, both modulo 65521 -- per-lane partial sums, added up by lane 0
constexpr uint32_t kNfiFlushBytes = kNfiWindow / 4;   // flushed when at least this much is new; a round never lets it pass the window
NFI_FN void nfi_flush(NfiCtx &c, uint8_t *dst, bool final)
{
    if (!final && c.pos - c.flushed < kNfiFlushBytes) return;     // uniform: both are read after a barrier
    const uint32_t from = c.flushed, upto = final ? c.pos : (c.pos & ~3u);
    const uint32_t n = upto - from;
    uint32_t pa = 0;
    uint64_t pb = 0;                            // 64 bits: the host build runs all of a flush (up to 8 KiB) on one "lane"
    if (((uintptr_t)dst & 3u) == 0) {           // `from` is a multiple of 4 (every earlier flush ended on one)
        const uint32_t nwords = n >> 2;
        uint32_t *d32 = reinterpret_cast<uint32_t *>(dst + from);
        NFI_FOR_LANES(w, nwords) {
            const uint32_t i0 = 4u * (uint32_t)w, p = (from + i0) & (kNfiWindow - 1);
            const uint32_t v = *reinterpret_cast<const uint32_t *>(&c.window[p]);
            d32[w] = v;
            const uint32_t b0 = v & 255u, b1 = (v >> 8) & 255u, b2 = (v >> 16) & 255u, b3 = v >> 24;
            const uint32_t s = b0 + b1 + b2 + b3;
            pa += s;
            pb += (uint64_t)((n - i0) * s - (b1 + 2u * b2 + 3u * b3));
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
    for (int o = 32; o > 0; o >>= 1) {          // wavefront sums (64 x 65520 and 64 x 8 KiB x 255 fit 32 bits)
        sa += __shfl_xor(sa, o, 64);
        sb += __shfl_xor(sb, o, 64);
    }
#endif
#ifndef NFI_HOST
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");      // far matches read these bytes back (other lanes' stores)
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
            nfi_decode_round(c, dst, out_len);     // all lanes, uniform
            NFI_SYNC();
            if (!c.err) nfi_flush(c, dst, false);  // uniform; a round that went wrong may have run past out_len
        } else {
            nfi_stored_round(c, words, nwords, out_len);
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
