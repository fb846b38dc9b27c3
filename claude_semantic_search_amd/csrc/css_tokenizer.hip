// css_tokenizer.hip -- host-side WordPiece front end of encode() (no device code in this file).
//
// The reference reaches its tokenizer through SentenceTransformer.encode (src/embeddings.py:184-188,
// :216-222), i.e. transformers' MPNetTokenizer on the native HF `tokenizers` library: BertNormalizer
// (clean text, isolate CJK, strip accents, lower-case) + BertPreTokenizer (whitespace, every punctuation
// mark alone) + greedy longest-match WordPiece ("##" continuation, 100-char word limit, [UNK]) +
// "<s> ... </s>" + truncation to max_seq_length (src/embeddings.py:97).  A 1.5 kB chunk costs the encoder
// ~0.1 ms of GPU time and the Python tokenizer ~2 ms, so the text path needs a native front end.
//
// This implementation handles texts that are pure ASCII (the bulk of code / chat transcripts) on all host
// cores; a text containing any byte >= 0x80 is reported back (len = -1) and tokenised by the Python
// implementation of the same pipeline (claude_semantic_search_amd/tokenizer.py), which is what pins the
// Unicode rules (NFD, general categories) against transformers in tests/test_tokenizer.py.
#include "css_common.h"

#include <algorithm>
#include <fstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

struct css_tokenizer {
    std::unordered_map<std::string, int32_t> vocab;
    int32_t unk = 3, bos = 0, eos = 2, pad = 1;
    bool lower = true;
    int max_chars = 100;
};

namespace {

inline bool ascii_punct(unsigned char c) {
    return (c >= 33 && c <= 47) || (c >= 58 && c <= 64) || (c >= 91 && c <= 96) || (c >= 123 && c <= 126);
}

// one pure-ASCII text -> ids (without specials), at most `budget` of them
void encode_ascii(const css_tokenizer& t, const char* s, int64_t n, int budget, std::vector<int32_t>& out,
                  std::string& word, std::string& key) {
    out.clear();
    word.clear();
    auto flush_word = [&]() {
        if (word.empty()) return;
        if ((int)word.size() > t.max_chars) {
            out.push_back(t.unk);
            word.clear();
            return;
        }
        const size_t first = out.size();
        size_t start = 0;
        const size_t L = word.size();
        bool bad = false;
        while (start < L) {
            size_t end = L;
            int32_t cur = -1;
            while (start < end) {
                key.clear();
                if (start > 0) key.append("##");
                key.append(word, start, end - start);
                auto it = t.vocab.find(key);
                if (it != t.vocab.end()) {
                    cur = it->second;
                    break;
                }
                --end;
            }
            if (cur < 0) {
                bad = true;
                break;
            }
            out.push_back(cur);
            start = end;
        }
        if (bad) {
            out.resize(first);
            out.push_back(t.unk);
        }
        word.clear();
    };
    for (int64_t i = 0; i < n && (int)out.size() < budget; ++i) {
        unsigned char c = (unsigned char)s[i];
        if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
            flush_word();
        } else if (c < 0x20 || c == 0x7F) {
            continue;  // control characters (incl. NUL) vanish without splitting the word
        } else if (ascii_punct(c)) {
            flush_word();
            key.assign(1, (char)c);
            auto it = t.vocab.find(key);
            out.push_back(it != t.vocab.end() ? it->second : t.unk);
        } else {
            if (t.lower && c >= 'A' && c <= 'Z') c = (unsigned char)(c + 32);
            word.push_back((char)c);
        }
    }
    if ((int)out.size() < budget) flush_word();
    if ((int)out.size() > budget) out.resize(budget);
}

}  // namespace

extern "C" {

int css_tokenizer_create(const char* vocab_path, int lowercase, css_tokenizer** out) {
    CSS_REQUIRE(vocab_path && out, "css_tokenizer_create: NULL argument");
    std::ifstream f(vocab_path, std::ios::binary);
    if (!f) {
        css::set_error("css_tokenizer_create: cannot open %s", vocab_path);
        return CSS_ERR_INVALID;
    }
    css_tokenizer* t = new css_tokenizer();
    t->lower = lowercase != 0;
    std::string line;
    int32_t id = 0;
    while (std::getline(f, line)) {
        if (!line.empty() && line.back() == '\r') line.pop_back();
        t->vocab[line] = id++;  // a duplicated line keeps its last id (as a Python dict built in file order does)
    }
    auto get = [&](const char* a, const char* b, int32_t dflt) {
        auto it = t->vocab.find(a);
        if (it != t->vocab.end()) return it->second;
        it = t->vocab.find(b);
        return it != t->vocab.end() ? it->second : dflt;
    };
    t->unk = get("[UNK]", "<unk>", 3);  // MPNet's tokenizer config names "[UNK]" as the unknown token
    t->bos = get("<s>", "[CLS]", 0);
    t->eos = get("</s>", "[SEP]", 2);
    t->pad = get("<pad>", "[PAD]", 1);
    *out = t;
    return CSS_OK;
}

int css_tokenizer_free(css_tokenizer* t) {
    delete t;
    return CSS_OK;
}

int css_tokenizer_vocab_size(const css_tokenizer* t, int* n) {
    CSS_REQUIRE(t && n, "css_tokenizer_vocab_size: NULL argument");
    *n = (int)t->vocab.size();
    return CSS_OK;
}

int css_tokenizer_encode_batch(const css_tokenizer* t, const char* bytes, const int64_t* offsets, int64_t n, int max_len,
                               int32_t* ids_out, int32_t* lens_out, int nthreads) {
    CSS_REQUIRE(t && offsets && ids_out && lens_out && (bytes || n == 0), "css_tokenizer_encode_batch: NULL argument");
    CSS_REQUIRE(n >= 0 && max_len >= 2, "css_tokenizer_encode_batch: bad sizes (n=%lld, max_len=%d)", (long long)n, max_len);
    if (n == 0) return CSS_OK;
    int nt = nthreads > 0 ? nthreads : (int)std::min<unsigned>(std::max(1u, std::thread::hardware_concurrency()), 32u);
    nt = (int)std::min<int64_t>(nt, n);
    auto work = [&](int64_t lo, int64_t hi) {
        std::vector<int32_t> ids;
        std::string word, key;
        for (int64_t i = lo; i < hi; ++i) {
            const char* s = bytes + offsets[i];
            const int64_t len = offsets[i + 1] - offsets[i];
            bool ascii = true;
            for (int64_t j = 0; j < len; ++j)
                if ((unsigned char)s[j] >= 0x80) {
                    ascii = false;
                    break;
                }
            int32_t* row = ids_out + (size_t)i * max_len;
            if (!ascii) {
                lens_out[i] = -1;  // the caller tokenises this text with the Unicode-complete implementation
                continue;
            }
            encode_ascii(*t, s, len, max_len - 2, ids, word, key);
            int p = 0;
            row[p++] = t->bos;
            for (int32_t v : ids) row[p++] = v;
            row[p++] = t->eos;
            lens_out[i] = p;
            for (; p < max_len; ++p) row[p] = t->pad;
        }
    };
    if (nt <= 1) {
        work(0, n);
    } else {
        std::vector<std::thread> th;
        const int64_t per = (n + nt - 1) / nt;
        for (int k = 0; k < nt; ++k) {
            const int64_t lo = k * per, hi = std::min<int64_t>(n, lo + per);
            if (lo < hi) th.emplace_back(work, lo, hi);
        }
        for (auto& x : th) x.join();
    }
    return CSS_OK;
}

}  // extern "C"
