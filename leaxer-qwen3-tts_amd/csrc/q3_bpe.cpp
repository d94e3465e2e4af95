// q3_bpe.cpp — see q3_bpe.h.  Written from the behaviour of reference src/io/tokenizer.cpp; the
// pre-tokenizer is a hand-written scanner (the reference builds a std::regex per call), the merge
// loop keeps ranks per symbol boundary instead of re-hashing every pair each round.
#include "q3_bpe.h"

#include <climits>
#include <cstdio>
#include <cstring>

namespace q3 {

// ---------------------------------------------------------------------------------------------
// alphabet (reference tokenizer.cpp:29-94): bytes 33..126, 161..172, 174..255 stand for themselves
// as ONE raw byte (not their UTF-8 form — this is where the reference departs from GPT-2); the other
// 68 bytes become U+0100 + (their index among those 68), UTF-8 encoded (always two bytes).
// ---------------------------------------------------------------------------------------------
static bool self_mapped(unsigned b) { return (b >= 33 && b <= 126) || (b >= 161 && b <= 172) || (b >= 174 && b <= 255); }

const std::string& BpeTokenizer::symbol(unsigned char b) {
    static const std::vector<std::string> table = [] {
        std::vector<std::string> t(256);
        unsigned shifted = 0;
        for (unsigned v = 0; v < 256; ++v) {
            if (self_mapped(v)) {
                t[v].assign(1, (char)v);
            } else {
                const unsigned cp = 0x100 + shifted++;
                t[v].push_back((char)(0xC0 | (cp >> 6)));
                t[v].push_back((char)(0x80 | (cp & 0x3F)));
            }
        }
        return t;
    }();
    return table[b];
}

// ---------------------------------------------------------------------------------------------
// vocab.json (reference tokenizer.cpp:103-285): a flat object of "string": non-negative-integer.
// Accepted looseness is part of the contract: commas are skipped wherever they appear, unknown
// escapes keep the escaped character, \uXXXX is encoded per 16-bit unit (no surrogate pairing),
// a repeated key keeps the LAST id, anything after the closing brace is ignored.
// ---------------------------------------------------------------------------------------------
static bool is_c_space(unsigned char c) { return c == ' ' || (c >= 9 && c <= 13); }
static int hex_val(char c) {
    if (c >= '0' && c <= '9') return c - '0';
    if (c >= 'a' && c <= 'f') return c - 'a' + 10;
    if (c >= 'A' && c <= 'F') return c - 'A' + 10;
    return -1;
}

bool BpeTokenizer::load_vocab(const std::string& path) {
    ids_.clear();
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) {
        fprintf(stderr, "Failed to open vocab file: %s\n", path.c_str());
        return false;
    }
    fseek(f, 0, SEEK_END);
    const long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    if (sz <= 0 || sz > 100L * 1024 * 1024) { // same bound as the reference (:120)
        fprintf(stderr, "Invalid file size: %ld\n", sz);
        fclose(f);
        return false;
    }
    std::string buf((size_t)sz, '\0');
    const size_t got = fread(&buf[0], 1, (size_t)sz, f);
    fclose(f);
    if (got != (size_t)sz) {
        fprintf(stderr, "Failed to read vocab file: %s\n", path.c_str());
        return false;
    }
    const char* s = buf.data();
    const size_t n = buf.size();
    size_t i = 0;
    auto skip_ws = [&] { while (i < n && is_c_space((unsigned char)s[i])) ++i; };
    auto fail = [&](const char* what) {
        fprintf(stderr, "vocab.json: %s (offset %zu)\n", what, i);
        return false; // entries read so far stay, as in the reference (the loaded flag is not set)
    };

    skip_ws();
    if (i >= n || s[i] != '{') return fail("expected '{'");
    ++i;
    std::string key;
    for (;;) {
        skip_ws();
        if (i >= n || s[i] == '}') break;
        if (s[i] == ',') { ++i; continue; }
        if (s[i] != '"') return fail("expected '\"'");
        ++i;
        key.clear();
        while (i < n && s[i] != '"') {
            char c = s[i];
            if (c != '\\') { key.push_back(c); ++i; continue; }
            if (++i >= n) return fail("escape at end of file");
            c = s[i];
            switch (c) {
            case 'n': key.push_back('\n'); ++i; break;
            case 't': key.push_back('\t'); ++i; break;
            case 'r': key.push_back('\r'); ++i; break;
            case 'u': {
                if (i + 4 >= n) return fail("truncated \\u escape");
                unsigned cp = 0;
                for (int k = 1; k <= 4; ++k) {
                    const int h = hex_val(s[i + k]);
                    if (h < 0) return fail("bad hex digit in \\u escape");
                    cp = cp << 4 | (unsigned)h;
                }
                if (cp < 0x80) {
                    key.push_back((char)cp);
                } else if (cp < 0x800) {
                    key.push_back((char)(0xC0 | (cp >> 6)));
                    key.push_back((char)(0x80 | (cp & 0x3F)));
                } else {
                    key.push_back((char)(0xE0 | (cp >> 12)));
                    key.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
                    key.push_back((char)(0x80 | (cp & 0x3F)));
                }
                i += 5;
                break;
            }
            default: key.push_back(c); ++i; break; // \\ \" \/ \b \f ...: the character itself
            }
        }
        if (i >= n) return fail("unterminated string");
        ++i;
        skip_ws();
        if (i >= n || s[i] != ':') return fail("expected ':'");
        ++i;
        skip_ws();
        if (i >= n || s[i] < '0' || s[i] > '9') return fail("expected a digit");
        uint32_t id = 0;
        while (i < n && s[i] >= '0' && s[i] <= '9') id = id * 10u + (uint32_t)(s[i++] - '0');
        ids_[key] = (int32_t)id;
    }
    if (ids_.empty()) return false;
    have_vocab_ = true;
    return true;
}

// ---------------------------------------------------------------------------------------------
// merges.txt (reference tokenizer.cpp:303-354): read in 1023-byte fgets units, CR/LF stripped, empty
// units skipped, split at the FIRST space, units without a space skipped.  No comment syntax: the
// "#version: 0.2" header of a Hugging Face merges file is an ordinary pair ("#version:", "0.2") and
// takes rank 0.  A repeated pair keeps its LAST rank.
// ---------------------------------------------------------------------------------------------
bool BpeTokenizer::load_merges(const std::string& path) {
    rank_.clear();
    n_merges_ = 0;
    FILE* f = fopen(path.c_str(), "r");
    if (!f) {
        fprintf(stderr, "Failed to open merges file: %s\n", path.c_str());
        return false;
    }
    char unit[1024];
    int next_rank = 0;
    std::string key;
    while (fgets(unit, sizeof unit, f)) {
        size_t n = strlen(unit);
        while (n && (unit[n - 1] == '\n' || unit[n - 1] == '\r')) unit[--n] = 0;
        if (!n) continue;
        const char* sp = (const char*)memchr(unit, ' ', n);
        if (!sp) {
            fprintf(stderr, "Invalid merge line (no space): %.60s\n", unit);
            continue;
        }
        key.assign(unit, (size_t)(sp - unit));
        key.push_back('\n');
        key.append(sp + 1, n - (size_t)(sp - unit) - 1);
        rank_[key] = next_rank++;
        ++n_merges_;
    }
    fclose(f);
    have_merges_ = true;
    return n_merges_ != 0;
}

// ---------------------------------------------------------------------------------------------
// pre-tokenizer (reference tokenizer.cpp:366-372).  The reference pattern, ECMAScript semantics
// (leftmost match, alternatives in order, greedy with backtracking, byte-wise, "C" locale):
//     's|'t|'re|'ve|'m|'ll|'d| ?[A-Za-z]+|[0-9]+| ?[^\s\w]+|\s+
// A byte no alternative can start on ('_' is the only one) is dropped.  Bytes >= 0x80 are neither
// \s nor \w, so they fall in the punctuation class.
// ---------------------------------------------------------------------------------------------
enum : unsigned char { C_LETTER = 1, C_DIGIT = 2, C_SPACE = 4, C_WORD = 8 };
static const unsigned char* char_classes() {
    static unsigned char t[256];
    static bool init = [] {
        for (int c = 0; c < 256; ++c) {
            unsigned char m = 0;
            if ((c >= 'A' && c <= 'Z') || (c >= 'a' && c <= 'z')) m |= C_LETTER | C_WORD;
            if (c >= '0' && c <= '9') m |= C_DIGIT | C_WORD;
            if (c == '_') m |= C_WORD;
            if (c == ' ' || (c >= 9 && c <= 13)) m |= C_SPACE;
            t[c] = m;
        }
        return true;
    }();
    (void)init;
    return t;
}

void BpeTokenizer::split(const char* text, size_t len, std::vector<std::pair<uint32_t, uint32_t>>& pieces) {
    const unsigned char* cls = char_classes();
    const unsigned char* s = (const unsigned char*)text;
    auto is = [&](size_t k, unsigned char m) { return k < len && (cls[s[k]] & m); };
    auto punct = [&](size_t k) { return k < len && !(cls[s[k]] & (C_SPACE | C_WORD)); };
    size_t i = 0;
    while (i < len) {
        size_t j = i;
        const unsigned char c = s[i];
        if (c == '\'' && i + 1 < len) { // contractions, in the pattern's order
            const unsigned char a = s[i + 1], b = i + 2 < len ? s[i + 2] : 0;
            if (a == 's' || a == 't') j = i + 2;
            else if ((a == 'r' || a == 'v') && b == 'e') j = i + 3;
            else if (a == 'm') j = i + 2;
            else if (a == 'l' && b == 'l') j = i + 3;
            else if (a == 'd') j = i + 2;
        }
        if (j == i) {
            const size_t w = (c == ' ') ? i + 1 : i; // optional leading space
            if (is(w, C_LETTER)) {
                j = w + 1;
                while (is(j, C_LETTER)) ++j;
            } else if (is(i, C_DIGIT)) {
                j = i + 1;
                while (is(j, C_DIGIT)) ++j;
            } else if (punct(w)) {
                j = w + 1;
                while (punct(j)) ++j;
            } else if (cls[c] & C_SPACE) {
                j = i + 1;
                while (is(j, C_SPACE)) ++j;
            }
        }
        if (j == i) { ++i; continue; } // unmatched byte: skipped by the regex iterator
        pieces.emplace_back((uint32_t)i, (uint32_t)j);
        i = j;
    }
}

// ---------------------------------------------------------------------------------------------
// merge loop (reference tokenizer.cpp:387-432): repeatedly join the adjacent pair of lowest rank,
// leftmost on ties, until no adjacent pair has a rank.
// ---------------------------------------------------------------------------------------------
void BpeTokenizer::merge_piece(const char* p, size_t n, std::vector<std::string>& sym) const {
    sym.clear();
    for (size_t k = 0; k < n; ++k) sym.push_back(symbol((unsigned char)p[k]));
    if (sym.size() < 2) return;
    std::string key;
    auto rank_of = [&](size_t k) { // rank of the boundary between sym[k] and sym[k+1]
        key.assign(sym[k]);
        key.push_back('\n');
        key.append(sym[k + 1]);
        const auto it = rank_.find(key);
        return it == rank_.end() ? INT_MAX : it->second;
    };
    std::vector<int> br(sym.size() - 1);
    for (size_t k = 0; k + 1 < sym.size(); ++k) br[k] = rank_of(k);
    while (!br.empty()) {
        size_t at = 0;
        int best = INT_MAX;
        for (size_t k = 0; k < br.size(); ++k)
            if (br[k] < best) { best = br[k]; at = k; }
        if (best == INT_MAX) break;
        sym[at] += sym[at + 1];
        sym.erase(sym.begin() + (long)at + 1);
        br.erase(br.begin() + (long)at);
        if (at > 0) br[at - 1] = rank_of(at - 1);
        if (at < br.size()) br[at] = rank_of(at);
    }
}

// ---------------------------------------------------------------------------------------------
// text -> ids (reference tokenizer.cpp:434-486).  Without a vocab: the raw bytes.  Without merges:
// each raw byte (not its alphabet symbol) is looked up.  A symbol missing from the vocab becomes the
// values of its bytes.
// ---------------------------------------------------------------------------------------------
void BpeTokenizer::encode(const char* text, size_t len, std::vector<int32_t>& out) const {
    out.clear();
    if (!len) return;
    if (!have_vocab_) {
        for (size_t k = 0; k < len; ++k) out.push_back((int32_t)(unsigned char)text[k]);
        return;
    }
    std::vector<std::pair<uint32_t, uint32_t>> pieces;
    split(text, len, pieces);
    std::vector<std::string> sym;
    for (const auto& pc : pieces) {
        if (have_merges_) {
            merge_piece(text + pc.first, pc.second - pc.first, sym);
        } else {
            sym.clear();
            for (uint32_t k = pc.first; k < pc.second; ++k) sym.emplace_back(1, text[k]);
        }
        for (const std::string& t : sym) {
            const auto it = ids_.find(t);
            if (it != ids_.end()) out.push_back(it->second);
            else
                for (unsigned char b : t) out.push_back((int32_t)b);
        }
    }
}

} // namespace q3
