// Minimal JSON reader/writer for the option structs that cross the C ABI
// (the text produced by Go's encoding/json for bigseqkit's option structs,
// /root/reference/bigseqkit/helper.go:47-66).  Header-only, host code.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace bsk {
namespace json {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool b = false;
    double num = 0;
    bool is_int = false;
    int64_t inum = 0;
    std::string str;
    std::vector<ValuePtr> arr;
    std::vector<std::pair<std::string, ValuePtr>> obj;  // keeps order

    const Value* get(const std::string& k) const {
        for (auto& kv : obj)
            if (kv.first == k) return kv.second.get();
        return nullptr;
    }
};

class Parser {
   public:
    explicit Parser(const std::string& s) : s_(s) {}
    ValuePtr parse() {
        skip();
        ValuePtr v = value();
        skip();
        if (p_ != s_.size()) fail("trailing characters");
        return v;
    }

   private:
    const std::string& s_;
    size_t p_ = 0;
    [[noreturn]] void fail(const char* m) {
        throw std::runtime_error(std::string("invalid options JSON: ") + m + " at offset " + std::to_string(p_));
    }
    void skip() {
        while (p_ < s_.size() && (s_[p_] == ' ' || s_[p_] == '\n' || s_[p_] == '\t' || s_[p_] == '\r')) ++p_;
    }
    bool lit(const char* w) {
        size_t n = strlen(w);
        if (s_.compare(p_, n, w) == 0) { p_ += n; return true; }
        return false;
    }
    static void put_utf8(std::string& o, unsigned cp) {
        if (cp < 0x80) o.push_back((char)cp);
        else if (cp < 0x800) { o.push_back((char)(0xC0 | (cp >> 6))); o.push_back((char)(0x80 | (cp & 0x3F))); }
        else if (cp < 0x10000) {
            o.push_back((char)(0xE0 | (cp >> 12)));
            o.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            o.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            o.push_back((char)(0xF0 | (cp >> 18)));
            o.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            o.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            o.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }
    unsigned hex4() {
        if (p_ + 4 > s_.size()) fail("bad \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; ++i) {
            char c = s_[p_++];
            v <<= 4;
            if (c >= '0' && c <= '9') v |= c - '0';
            else if (c >= 'a' && c <= 'f') v |= c - 'a' + 10;
            else if (c >= 'A' && c <= 'F') v |= c - 'A' + 10;
            else fail("bad \\u escape");
        }
        return v;
    }
    std::string string() {
        if (s_[p_] != '"') fail("expected string");
        ++p_;
        std::string o;
        while (p_ < s_.size() && s_[p_] != '"') {
            char c = s_[p_++];
            if (c != '\\') { o.push_back(c); continue; }
            if (p_ >= s_.size()) fail("bad escape");
            char e = s_[p_++];
            switch (e) {
                case '"': o.push_back('"'); break;
                case '\\': o.push_back('\\'); break;
                case '/': o.push_back('/'); break;
                case 'b': o.push_back('\b'); break;
                case 'f': o.push_back('\f'); break;
                case 'n': o.push_back('\n'); break;
                case 'r': o.push_back('\r'); break;
                case 't': o.push_back('\t'); break;
                case 'u': {
                    unsigned cp = hex4();
                    if (cp >= 0xD800 && cp < 0xDC00 && p_ + 1 < s_.size() && s_[p_] == '\\' && s_[p_ + 1] == 'u') {
                        p_ += 2;
                        unsigned lo = hex4();
                        cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                    }
                    put_utf8(o, cp);
                    break;
                }
                default: fail("bad escape");
            }
        }
        if (p_ >= s_.size()) fail("unterminated string");
        ++p_;
        return o;
    }
    ValuePtr value() {
        skip();
        if (p_ >= s_.size()) fail("unexpected end");
        auto v = std::make_shared<Value>();
        char c = s_[p_];
        if (c == '{') {
            v->kind = Value::Object;
            ++p_;
            skip();
            if (p_ < s_.size() && s_[p_] == '}') { ++p_; return v; }
            for (;;) {
                skip();
                std::string k = string();
                skip();
                if (p_ >= s_.size() || s_[p_] != ':') fail("expected ':'");
                ++p_;
                v->obj.emplace_back(k, value());
                skip();
                if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
                if (p_ < s_.size() && s_[p_] == '}') { ++p_; break; }
                fail("expected ',' or '}'");
            }
        } else if (c == '[') {
            v->kind = Value::Array;
            ++p_;
            skip();
            if (p_ < s_.size() && s_[p_] == ']') { ++p_; return v; }
            for (;;) {
                v->arr.push_back(value());
                skip();
                if (p_ < s_.size() && s_[p_] == ',') { ++p_; continue; }
                if (p_ < s_.size() && s_[p_] == ']') { ++p_; break; }
                fail("expected ',' or ']'");
            }
        } else if (c == '"') {
            v->kind = Value::String;
            v->str = string();
        } else if (lit("true")) {
            v->kind = Value::Bool;
            v->b = true;
        } else if (lit("false")) {
            v->kind = Value::Bool;
            v->b = false;
        } else if (lit("null")) {
            v->kind = Value::Null;
        } else {
            size_t st = p_;
            bool isint = true;
            if (p_ < s_.size() && (s_[p_] == '-' || s_[p_] == '+')) ++p_;
            while (p_ < s_.size() && ((s_[p_] >= '0' && s_[p_] <= '9') || s_[p_] == '.' || s_[p_] == 'e' ||
                                      s_[p_] == 'E' || s_[p_] == '-' || s_[p_] == '+')) {
                if (s_[p_] == '.' || s_[p_] == 'e' || s_[p_] == 'E') isint = false;
                ++p_;
            }
            if (p_ == st) fail("unexpected character");
            std::string t = s_.substr(st, p_ - st);
            v->kind = Value::Number;
            v->num = strtod(t.c_str(), nullptr);
            v->is_int = isint;
            v->inum = isint ? strtoll(t.c_str(), nullptr, 10) : (int64_t)v->num;
        }
        return v;
    }
};

// Go's encoding/json string escaping (HTML-safe: <, >, & as \u00XX)
inline std::string quote(const std::string& s) {
    std::string o = "\"";
    char b[8];
    for (unsigned char c : s) {
        switch (c) {
            case '"': o += "\\\""; break;
            case '\\': o += "\\\\"; break;
            case '\n': o += "\\n"; break;
            case '\r': o += "\\r"; break;
            case '\t': o += "\\t"; break;
            case '<': case '>': case '&':
                snprintf(b, sizeof b, "\\u%04x", c);
                o += b;
                break;
            default:
                if (c < 0x20) { snprintf(b, sizeof b, "\\u%04x", c); o += b; }
                else o.push_back((char)c);
        }
    }
    o.push_back('"');
    return o;
}

}  // namespace json
}  // namespace bsk
