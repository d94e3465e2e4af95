// q3_bpe.h — byte-level BPE text tokenizer for the prompt side of the path.
// Behavioural contract = the reference's leaxer_qwen::io tokenizer (reference src/io/tokenizer.h:13-28,
// src/io/tokenizer.cpp:29-94 byte alphabet, :103-285 vocab.json reader, :303-354 merges reader,
// :357-384 pre-tokenizer, :387-432 merge loop, :434-486 id lookup + byte fallback), including its
// deviations from the Hugging Face Qwen2 tokenizer — the ids must equal the reference's, not HF's.
// Host-only code (no GPU work): the ids feed q3tts_build_prompt_host.
#ifndef Q3_BPE_H
#define Q3_BPE_H

#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>

namespace q3 {

class BpeTokenizer {
public:
    bool load_vocab(const std::string& path);  // flat {"token": id, ...} JSON
    bool load_merges(const std::string& path); // one "left right" pair per line, rank = line order
    bool ready() const { return have_vocab_ && have_merges_; }
    size_t vocab_size() const { return ids_.size(); }
    size_t merges_size() const { return n_merges_; }

    void encode(const char* text, size_t len, std::vector<int32_t>& out) const;
    // the pre-tokenizer alone (pieces as [begin,end) byte ranges of `text`); exposed for the tests
    static void split(const char* text, size_t len, std::vector<std::pair<uint32_t, uint32_t>>& pieces);
    // alphabet symbol (1 or 2 bytes) for one input byte
    static const std::string& symbol(unsigned char b);

private:
    void merge_piece(const char* p, size_t n, std::vector<std::string>& sym) const;
    bool have_vocab_ = false, have_merges_ = false;
    size_t n_merges_ = 0;
    std::unordered_map<std::string, int32_t> ids_;
    std::unordered_map<std::string, int> rank_; // key: left + '\n' + right ('\n' cannot occur inside a merges line)
};

} // namespace q3
#endif
