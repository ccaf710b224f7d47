// l3_rules.h -- the pre-tokenizer as POSITION-LOCAL rules (host- and device-compilable).
//
// The reference walks the text sequentially with six ordered matchers
// (llama3/internal/pretokenizer/state_machine.go:82-284) after splitting on special tokens
// (internal/tokens/special.go:12,48-83).  On the GPU every byte decides for itself whether a
// pre-token starts at it, from: its own rune, at most 4 runes of look-behind, 1 rune of
// look-ahead and the offset inside a digit run.  These functions are the single
// statement of those rules: the tile kernel calls them on LDS-resident bytes, the long-pre-token
// kernel on global memory, and tests/test_rules_host.py fuzzes the host build against the oracle.
#pragma once
#include "l3_common.h"

namespace l3 {

// ---- per-byte info word ------------------------------------------------------------------------
enum : uint32_t { C_P = 0, C_L = 1, C_N = 2, C_NL = 3, C_SP = 4, C_WS = 5 };  // class (bits 0..2)
constexpr uint32_t I_CLS = 0x7;
constexpr uint32_t I_RUNE = 0x8;       // a rune (or a special token) starts at this byte
constexpr uint32_t I_CC_SHIFT = 4;     // bits 4..7: contraction code of an ASCII rune
constexpr uint32_t I_INVALID = 0x100;  // invalid UTF-8 byte: rune U+FFFD, re-encoded as EF BF BD
constexpr uint32_t I_SEG = 0x200;      // segment start: no look-behind across this byte
constexpr uint32_t I_SPECIAL = 0x400;  // first byte of a special token
constexpr uint32_t I_SPBODY = 0x800;   // inside a special token
enum : uint32_t { CC_NONE = 0, CC_APOS, CC_S, CC_T, CC_M, CC_D, CC_R, CC_E, CC_V, CC_L };

L3_HD bool is_ws_cls(uint32_t c) { return c >= C_NL; }
L3_HD bool is_cont(uint32_t b) { return (b & 0xC0u) == 0x80u; }

// Byte source concept `A`:
//   uint32_t b(int64_t q)   byte at q (q is inside the readable window)
//   bool     ds(int64_t q)  a document starts at q, or q is the end of all text
//   int64_t  lo(), hi()     readable window [lo, hi): bytes outside are treated as another document

// byte at p+k if it lies in the same document (and window) as p, else 0x100
template <class A>
L3_HD uint32_t fwd_byte(const A& a, int64_t p, int k) {
    for (int j = 1; j <= k; j++)
        if (p + j >= a.hi() || a.ds(p + j)) return 0x100u;
    return a.b(p + k);
}

// Length of the valid UTF-8 sequence starting at q (Go utf8.DecodeRune acceptance);
// 1 for ASCII, 0 if the byte at q does not start a valid sequence.
template <class A>
L3_HD int seq_len(const A& a, int64_t q) {
    uint32_t b0 = a.b(q);
    if (b0 < 0x80u) return 1;
    if (b0 < 0xC2u || b0 > 0xF4u) return 0;
    uint32_t b1 = fwd_byte(a, q, 1);
    if (b0 < 0xE0u) return is_cont(b1) ? 2 : 0;
    if (b0 < 0xF0u) {
        uint32_t lo = (b0 == 0xE0u) ? 0xA0u : 0x80u, hi = (b0 == 0xEDu) ? 0x9Fu : 0xBFu;
        if (b1 < lo || b1 > hi) return 0;
        return is_cont(fwd_byte(a, q, 2)) ? 3 : 0;
    }
    uint32_t lo = (b0 == 0xF0u) ? 0x90u : 0x80u, hi = (b0 == 0xF4u) ? 0x8Fu : 0xBFu;
    if (b1 < lo || b1 > hi) return 0;
    if (!is_cont(fwd_byte(a, q, 2))) return 0;
    return is_cont(fwd_byte(a, q, 3)) ? 4 : 0;
}

// seq_len() of a lead byte (>= 0xC0) given with the three bytes behind it as one little-endian word, for a lead whose
// next three bytes are known to lie in the same document and window: no document checks, no loop (the tile kernel's
// threads that see no document start near their bytes).
L3_HD uint32_t seq_len_word(uint32_t w) {
    const uint32_t b0 = w & 0xFFu, b1 = (w >> 8) & 0xFFu;
    const bool c1 = (w & 0xC000u) == 0x8000u, c2 = (w & 0xC00000u) == 0x800000u, c3 = (w & 0xC0000000u) == 0x80000000u;
    if (b0 < 0xC2u || b0 > 0xF4u) return 0;
    if (b0 < 0xE0u) return c1 ? 2u : 0u;
    if (b0 < 0xF0u) {
        const uint32_t lo = (b0 == 0xE0u) ? 0xA0u : 0x80u, hi = (b0 == 0xEDu) ? 0x9Fu : 0xBFu;
        return (b1 >= lo && b1 <= hi && c2) ? 3u : 0u;
    }
    const uint32_t lo = (b0 == 0xF0u) ? 0x90u : 0x80u, hi = (b0 == 0xF4u) ? 0x8Fu : 0xBFu;
    return (b1 >= lo && b1 <= hi && c2 && c3) ? 4u : 0u;
}

// Does a rune start at byte p?  Lead and ASCII bytes always do; a continuation byte does unless a
// valid sequence that started 1..3 bytes earlier in the same document covers it.
template <class A>
L3_HD bool rune_starts_at(const A& a, int64_t p) {
    if (!is_cont(a.b(p))) return true;
    for (int k = 1; k <= 3; k++) {
        if (a.ds(p - k + 1)) return true;
        int64_t q = p - k;
        if (q < a.lo()) return true;
        if (!is_cont(a.b(q))) return !(seq_len(a, q) > k);
    }
    return true;
}

// Blocks that are letters from end to end in Unicode 15.0 -- CJK Unified Ideographs U+4E00-9FFF, extension A U+3400-4DBF,
// Hangul syllables U+AC00-D7A3 (tests/test_unicode_tables.py checks them against the table): the bulk of CJK text gets
// its class without the two dependent table loads of the others
L3_HD bool letter_block(uint32_t cp) { return cp - 0x4E00u < 0x5200u || cp - 0x3400u < 0x19C0u || cp - 0xAC00u < 0x2BA4u; }
// the class of cp >= 0x80 out of the byte of its block that holds it: blocks[stage1[cp >> 8] * 64 + ((cp & 0xFF) >> 2)]
L3_HD uint32_t class_block_index(uint32_t blk, uint32_t cp) { return blk * 64u + ((cp & 0xFFu) >> 2); }
L3_HD uint32_t class_from_block_byte(uint32_t byte, uint32_t cp) {
    const uint32_t c = (byte >> ((cp & 3u) * 2u)) & 3u;
    return c == 3u ? (uint32_t)C_WS : c;  // non-ASCII white space is never ' ', '\r' or '\n'
}
L3_HD uint32_t class_of_cp(uint32_t cp, const uint16_t* stage1, const uint8_t* blocks) {
    if (cp < 0x80u) {  // ASCII without touching the table
        if ((cp | 0x20u) - 'a' < 26u) return C_L;
        if (cp - '0' < 10u) return C_N;
        if (cp == ' ') return C_SP;
        if (cp == '\n' || cp == '\r') return C_NL;
        if (cp - 9u < 5u) return C_WS;  // \t \v \f (and \n \r handled above)
        return C_P;
    }
    if (letter_block(cp)) return C_L;
    return class_from_block_byte(blocks[class_block_index(stage1[cp >> 8], cp)], cp);
}

L3_HD uint32_t contraction_code(uint32_t b) {
    switch (b | 0x20u) {
        case 's': return CC_S;
        case 't': return CC_T;
        case 'm': return CC_M;
        case 'd': return CC_D;
        case 'r': return CC_R;
        case 'e': return CC_E;
        case 'v': return CC_V;
        case 'l': return CC_L;
    }
    return b == '\'' ? (uint32_t)CC_APOS : (uint32_t)CC_NONE;
}

// Info word of byte p, special tokens NOT considered (caller overlays I_SPECIAL/I_SPBODY/I_SEG).
template <class A>
L3_HD uint32_t byte_info(const A& a, int64_t p, const uint16_t* stage1, const uint8_t* blocks) {
    uint32_t b0 = a.b(p);
    uint32_t seg = a.ds(p) ? I_SEG : 0u;
    if (b0 < 0x80u) return class_of_cp(b0, stage1, blocks) | I_RUNE | (contraction_code(b0) << I_CC_SHIFT) | seg;
    if (is_cont(b0)) return rune_starts_at(a, p) ? (C_P | I_RUNE | I_INVALID | seg) : 0u;
    int n = seq_len(a, p);
    if (n == 0) return C_P | I_RUNE | I_INVALID | seg;
    uint32_t cp;
    if (n == 2) cp = ((b0 & 0x1Fu) << 6) | (a.b(p + 1) & 0x3Fu);
    else if (n == 3) cp = ((b0 & 0x0Fu) << 12) | ((a.b(p + 1) & 0x3Fu) << 6) | (a.b(p + 2) & 0x3Fu);
    else cp = ((b0 & 0x07u) << 18) | ((a.b(p + 1) & 0x3Fu) << 12) | ((a.b(p + 2) & 0x3Fu) << 6) | (a.b(p + 3) & 0x3Fu);
    return class_of_cp(cp, stage1, blocks) | I_RUNE | seg;
}

// ---- special tokens (special.go:12): literal match at p, inside one document -------------------
// Returns the match length (0 = none) and the canonical index (l3_common.h) in *canon.
template <class A>
L3_HD int special_match(const A& a, int64_t p, int* canon) {
    if (a.b(p) != '<' || fwd_byte(a, p, 1) != '|') return 0;
    // bytes available in this document after p, up to the longest special token
    int avail = 2;
    while (avail < kMaxSpecialLen && fwd_byte(a, p + avail - 1, 1) != 0x100u) avail++;
    const char* names[kSpecialNamed] = {"begin_of_text", "end_of_text", "start_header_id", "end_header_id",
                                        "eot_id", "eom_id", "python_tag", "finetune_right_pad_id"};
    const int lens[kSpecialNamed] = {13, 11, 15, 13, 6, 6, 10, 21};
    for (int i = 0; i < kSpecialNamed; i++) {
        int l = lens[i];
        if (avail < l + 4) continue;
        bool ok = true;
        for (int k = 0; k < l && ok; k++) ok = a.b(p + 2 + k) == (uint32_t)(uint8_t)names[i][k];
        if (ok && a.b(p + 2 + l) == '|' && a.b(p + 3 + l) == '>') { *canon = i; return l + 4; }
    }
    const char* pre = "reserved_special_token_";
    const int pl = 23;
    if (avail < 2 + pl + 3) return 0;
    for (int k = 0; k < pl; k++)
        if (a.b(p + 2 + k) != (uint32_t)(uint8_t)pre[k]) return 0;
    int64_t d = p + 2 + pl;
    int room = avail - 2 - pl;  // bytes available from d on
    int v = 0;
    for (int dl = 1; dl <= 3; dl++) {
        if (room < dl + 2) return 0;
        uint32_t c = a.b(d + dl - 1);
        if (c - '0' >= 10u) return 0;
        v = v * 10 + (int)(c - '0');
        bool ok = !(dl > 1 && a.b(d) == '0') && v <= 247;
        if (ok && a.b(d + dl) == '|' && a.b(d + dl + 1) == '>') { *canon = kSpecialNamed + v; return 2 + pl + dl + 2; }
    }
    return 0;
}

// ---- start-of-pre-token predicate ---------------------------------------------------------------
// Info source concept `I`:
//   uint32_t info(int64_t q)   info word of byte q (with the special-token overlay applied)
//   int64_t  lo(), hi()        positions with a valid info word: [lo, hi)
// Navigation helpers return -1 when there is no such rune in the same segment / window.
template <class I>
L3_HD int64_t prev_rune(const I& s, int64_t q) {
    if (s.info(q) & I_SEG) return -1;
    int64_t r = q - 1;
    while (r >= s.lo() && !(s.info(r) & I_RUNE)) r--;
    return r >= s.lo() ? r : -2;  // -2: ran out of window (caller treats as unknown)
}
template <class I>
L3_HD int64_t next_rune(const I& s, int64_t q) {
    int64_t r = q + 1;
    while (r < s.hi() && !(s.info(r) & I_RUNE)) r++;
    if (r >= s.hi() || (s.info(r) & I_SEG)) return -1;
    return r;
}
L3_HD uint32_t cc_of(uint32_t info) { return (info >> I_CC_SHIFT) & 0xFu; }

// ' that begins a contraction candidate: at a segment start or after L / N / newline / other white space
template <class I>
L3_HD bool apostrophe_starts(const I& s, int64_t q) {
    uint32_t w = s.info(q);
    if (cc_of(w) != CC_APOS) return false;
    int64_t r = prev_rune(s, q);
    if (r < 0) return r == -1;
    uint32_t c = s.info(r) & I_CLS;
    return c == C_L || c == C_N || c == C_NL || c == C_WS;
}

// Result of the digit-run query: number of consecutive N runes immediately before q in the same
// segment, or -1 if the run reaches the lower edge of the window (caller must look further back).
template <class I>
L3_HD int digits_before(const I& s, int64_t q) {
    int n = 0;
    for (;;) {
        int64_t r = prev_rune(s, q);
        if (r == -2) return -1;
        if (r == -1 || (s.info(r) & I_CLS) != C_N) return n;
        n++; q = r;
    }
}

// Is byte p (which must have I_RUNE set and not be inside a special token) the first byte of a
// pre-token?  `digit_off` is digits_before(p) (only read when p and its predecessor are digits).
template <class I>
L3_HD bool starts_pretoken(const I& s, int64_t p, int digit_off) {
    uint32_t w = s.info(p);
    if (w & (I_SEG | I_SPECIAL)) return true;
    int64_t p1 = prev_rune(s, p);
    if (p1 < 0) return true;  // window exhausted: cannot happen with the halos the callers use
    uint32_t c = w & I_CLS, pc = s.info(p1) & I_CLS;
    switch (c) {
        case C_L: {
            if (pc == C_N || pc == C_NL) return true;
            if (pc == C_SP || pc == C_WS) return false;
            int64_t p2 = prev_rune(s, p1);
            if (pc == C_P) {
                if (p2 < 0) return false;
                uint32_t c2 = s.info(p2) & I_CLS;
                return c2 == C_P || c2 == C_SP;
            }
            // pc == L: only the end of a contraction ('s 't 'm 'd | 're 've 'll) splits a letter run
            if (p2 < 0) return false;
            uint32_t cc1 = cc_of(s.info(p1));
            if (cc_of(s.info(p2)) == CC_APOS)
                return (cc1 == CC_S || cc1 == CC_T || cc1 == CC_M || cc1 == CC_D) && apostrophe_starts(s, p2);
            int64_t p3 = prev_rune(s, p2);
            if (p3 < 0 || cc_of(s.info(p3)) != CC_APOS) return false;
            uint32_t cc2 = cc_of(s.info(p2));
            bool three = (cc2 == CC_R && cc1 == CC_E) || (cc2 == CC_V && cc1 == CC_E) || (cc2 == CC_L && cc1 == CC_L);
            return three && apostrophe_starts(s, p3);
        }
        case C_N:
            return pc != C_N || (digit_off % 3) == 0;
        case C_P:
            return !(pc == C_P || pc == C_SP);
        default: {  // white space
            if (pc == C_L || pc == C_N) return true;
            if (pc == C_P) return c != C_NL;
            if (c == C_NL) return false;
            if (pc == C_NL) return true;
            int64_t nx = next_rune(s, p);
            return nx >= 0 && !is_ws_cls(s.info(nx) & I_CLS);
        }
    }
}

}  // namespace l3
