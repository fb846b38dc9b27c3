// css_tokenizer.hip -- host-side WordPiece front end of encode() (no device code in this file).
//
// The reference reaches its tokenizer through SentenceTransformer.encode (src/embeddings.py:184-188,
// :216-222), i.e. transformers' MPNetTokenizer on the native HF `tokenizers` library: BertNormalizer
// (clean text, isolate CJK, strip accents, lower-case) + BertPreTokenizer (whitespace, every punctuation
// mark alone) + greedy longest-match WordPiece ("##" continuation, 100-char word limit, [UNK]) +
// "<s> ... </s>" + truncation to max_seq_length (src/embeddings.py:97).  A 1.5 kB chunk costs the encoder
// ~0.1 ms of GPU time and the Python tokenizer ~2 ms, so the text path needs a native front end.
//
// Texts are tokenised on all host cores.  Pure-ASCII texts (the bulk of code / chat transcripts) take a byte
// loop; other texts are decoded from UTF-8 and pushed through per-code-point tables generated from Python's
// unicodedata (css_unicode_tables.h, tools/gen_unicode_tables.py: dropped / separator / punctuation / CJK /
// NFD-stripped-lower-cased sequence), Hangul syllables are decomposed algorithmically.  What the tables cannot
// express is reported back (len = -1) and tokenised by the Python implementation of the same pipeline
// (claude_semantic_search_amd/tokenizer.py): invalid UTF-8, a capital sigma (str.lower()'s final-sigma rule needs
// context), and any non-ASCII text when lower-casing is off.  tests/test_tokenizer.py checks every code point.
#include "css_common.h"
#include "css_unicode_tables.h"

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

// one UTF-8 text (lower-casing pipeline) -> ids; false: the text needs the Python implementation
bool encode_utf8(const css_tokenizer& t, const char* s, int64_t n, int budget, std::vector<int32_t>& out,
                 std::string& word, std::vector<int>& offs, std::string& key) {
    using namespace css_uni;
    out.clear();
    word.clear();
    offs.clear();
    auto push_char = [&](uint32_t cp) {
        offs.push_back((int)word.size());
        if (cp < 0x80) {
            word.push_back((char)cp);
        } else if (cp < 0x800) {
            word.push_back((char)(0xC0 | (cp >> 6)));
            word.push_back((char)(0x80 | (cp & 63)));
        } else if (cp < 0x10000) {
            word.push_back((char)(0xE0 | (cp >> 12)));
            word.push_back((char)(0x80 | ((cp >> 6) & 63)));
            word.push_back((char)(0x80 | (cp & 63)));
        } else {
            word.push_back((char)(0xF0 | (cp >> 18)));
            word.push_back((char)(0x80 | ((cp >> 12) & 63)));
            word.push_back((char)(0x80 | ((cp >> 6) & 63)));
            word.push_back((char)(0x80 | (cp & 63)));
        }
    };
    auto flush_word = [&]() {
        const int nch = (int)offs.size();
        if (nch == 0) return;
        if (nch > t.max_chars) {
            out.push_back(t.unk);
        } else {
            offs.push_back((int)word.size());
            const size_t first = out.size();
            int start = 0;
            bool bad = false;
            while (start < nch) {
                int end = nch;
                int32_t cur = -1;
                while (start < end) {
                    key.clear();
                    if (start > 0) key.append("##");
                    key.append(word, offs[start], offs[end] - offs[start]);
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
        }
        word.clear();
        offs.clear();
    };
    auto emit_punct = [&](uint32_t cp) {
        flush_word();
        push_char(cp);
        key.assign(word);
        word.clear();
        offs.clear();
        auto it = t.vocab.find(key);
        out.push_back(it != t.vocab.end() ? it->second : t.unk);
    };
    int64_t i = 0;
    while (i < n && (int)out.size() < budget) {
        uint32_t cp;
        const unsigned char c0 = (unsigned char)s[i];
        if (c0 < 0x80) {
            cp = c0;
            i += 1;
        } else if (c0 >= 0xC2 && c0 <= 0xDF && i + 1 < n && ((unsigned char)s[i + 1] & 0xC0) == 0x80) {
            cp = ((c0 & 0x1F) << 6) | ((unsigned char)s[i + 1] & 0x3F);
            i += 2;
        } else if (c0 >= 0xE0 && c0 <= 0xEF && i + 2 < n && ((unsigned char)s[i + 1] & 0xC0) == 0x80 &&
                   ((unsigned char)s[i + 2] & 0xC0) == 0x80) {
            cp = ((c0 & 0x0F) << 12) | (((unsigned char)s[i + 1] & 0x3F) << 6) | ((unsigned char)s[i + 2] & 0x3F);
            if (cp < 0x800 || (cp >= 0xD800 && cp <= 0xDFFF)) return false;  // overlong / surrogate: not valid UTF-8
            i += 3;
        } else if (c0 >= 0xF0 && c0 <= 0xF4 && i + 3 < n && ((unsigned char)s[i + 1] & 0xC0) == 0x80 &&
                   ((unsigned char)s[i + 2] & 0xC0) == 0x80 && ((unsigned char)s[i + 3] & 0xC0) == 0x80) {
            cp = ((c0 & 0x07) << 18) | (((unsigned char)s[i + 1] & 0x3F) << 12) | (((unsigned char)s[i + 2] & 0x3F) << 6) |
                 ((unsigned char)s[i + 3] & 0x3F);
            if (cp < 0x10000 || cp > 0x10FFFF) return false;
            i += 4;
        } else {
            return false;
        }
        if (cp < 0x80) {  // the ASCII rules of encode_ascii
            if (cp == ' ' || cp == '\t' || cp == '\n' || cp == '\r') flush_word();
            else if (cp < 0x20 || cp == 0x7F) continue;
            else if (ascii_punct((unsigned char)cp)) emit_punct(cp);
            else push_char(cp >= 'A' && cp <= 'Z' ? cp + 32 : cp);
            continue;
        }
        if (cp == 0x03A3) return false;  // capital sigma: lower() depends on the position in the word
        switch (kPages[kPageIndex[cp >> 8]][cp & 255]) {
            case C_CHAR: push_char(cp); break;
            case C_DROP: break;
            case C_SPACE: flush_word(); break;
            case C_PUNCT: emit_punct(cp); break;
            case C_CJK:
                flush_word();
                push_char(cp);
                flush_word();
                break;
            case C_HANGUL: {
                const uint32_t si = cp - 0xAC00;
                push_char(0x1100 + si / 588);
                push_char(0x1161 + (si % 588) / 28);
                if (si % 28) push_char(0x11A7 + si % 28);
                break;
            }
            default: {  // C_MAP
                const uint32_t* kb = kMapKeys;
                const uint32_t* it = std::lower_bound(kb, kb + kMapCount, cp);
                if (it == kb + kMapCount || *it != cp) return false;  // (cannot happen: the class says it is mapped)
                const int idx = (int)(it - kb);
                for (int j = 0; j < kMapLen[idx]; ++j) {
                    const uint32_t e = kMapPool[kMapOff[idx] + j], kind = e >> 30, ocp = e & 0x3FFFFFFFu;
                    if (kind == K_CHAR) push_char(ocp);
                    else if (kind == K_SPACE) flush_word();
                    else emit_punct(ocp);
                }
            }
        }
    }
    if ((int)out.size() < budget) flush_word();
    if ((int)out.size() > budget) out.resize(budget);
    return true;
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
        std::vector<int> offs;
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
            if (ascii) {
                encode_ascii(*t, s, len, max_len - 2, ids, word, key);
            } else if (!t->lower || !encode_utf8(*t, s, len, max_len - 2, ids, word, offs, key)) {
                lens_out[i] = -1;  // the caller tokenises this text with the Python implementation
                continue;
            }
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
