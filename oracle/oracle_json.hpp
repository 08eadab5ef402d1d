// TEST INFRASTRUCTURE ONLY — part of the CPU oracle; nothing under imagecaptioner_amd/ may include it.
//
// Minimal JSON document model with the System.Text.Json behaviours the
// reference's loaders rely on (Program.cs:923, 1018, 1277, 1309, 1376):
//  * strict RFC 8259: no comments, no trailing commas, depth limit 64;
//  * object members keep document order, duplicates are kept;
//  * TryGetProperty on a non-object throws, and returns the LAST duplicate;
//  * GetString on a null element gives null (callers then apply `?? ""`),
//    on any other non-string kind it throws.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace orc {

typedef std::u16string S;

struct JsonError : std::runtime_error {
    explicit JsonError(const std::string& m) : std::runtime_error(m) {}
};

struct JV {
    enum Kind { Null, False, True, Number, String, Array, Object } kind = Null;
    S str;
    std::vector<JV> arr;
    std::vector<std::pair<S, JV>> obj;

    bool is_str() const { return kind == String; }
    bool is_arr() const { return kind == Array; }
    bool is_obj() const { return kind == Object; }
    // JsonElement.TryGetProperty: InvalidOperationException unless Object; last duplicate wins.
    const JV* get(const char* name) const {
        if (kind != Object) throw JsonError("TryGetProperty on a non-object element");
        S n;
        for (const char* p = name; *p; ++p) n.push_back((char16_t)(unsigned char)*p);
        for (size_t i = obj.size(); i-- > 0;)
            if (obj[i].first == n) return &obj[i].second;
        return nullptr;
    }
    // JsonElement.GetString() ?? "": null -> "", string -> value, anything else throws.
    S get_string_or_empty() const {
        if (kind == String) return str;
        if (kind == Null) return S();
        throw JsonError("GetString on a non-string element");
    }
};

class JsonParser {
public:
    JsonParser(const uint8_t* p, size_t n) : p_(p), n_(n), i_(0) {}
    JV parse_document() {
        if (n_ >= 3 && p_[0] == 0xEF && p_[1] == 0xBB && p_[2] == 0xBF) i_ = 3;  // File.ReadAllText drops the BOM
        skip_ws();
        JV v = parse_value(0);
        skip_ws();
        if (i_ != n_) fail("trailing characters after the JSON value");
        return v;
    }

private:
    const uint8_t* p_;
    size_t n_, i_;

    [[noreturn]] void fail(const char* m) const { throw JsonError(std::string(m) + " at byte " + std::to_string(i_)); }
    void skip_ws() {
        while (i_ < n_ && (p_[i_] == ' ' || p_[i_] == '\t' || p_[i_] == '\n' || p_[i_] == '\r')) ++i_;
    }
    bool lit(const char* s) {
        size_t k = 0;
        while (s[k]) {
            if (i_ + k >= n_ || p_[i_ + k] != (uint8_t)s[k]) return false;
            ++k;
        }
        i_ += k;
        return true;
    }
    JV parse_value(int depth) {
        if (depth > 64) fail("maximum depth exceeded");
        if (i_ >= n_) fail("unexpected end");
        JV v;
        uint8_t c = p_[i_];
        if (c == '{') {
            ++i_;
            v.kind = JV::Object;
            skip_ws();
            if (i_ < n_ && p_[i_] == '}') { ++i_; return v; }
            for (;;) {
                skip_ws();
                if (i_ >= n_ || p_[i_] != '"') fail("expected property name");
                S key = parse_string();
                skip_ws();
                if (i_ >= n_ || p_[i_] != ':') fail("expected ':'");
                ++i_;
                skip_ws();
                JV child = parse_value(depth + 1);
                v.obj.emplace_back(std::move(key), std::move(child));
                skip_ws();
                if (i_ < n_ && p_[i_] == ',') { ++i_; continue; }
                if (i_ < n_ && p_[i_] == '}') { ++i_; return v; }
                fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++i_;
            v.kind = JV::Array;
            skip_ws();
            if (i_ < n_ && p_[i_] == ']') { ++i_; return v; }
            for (;;) {
                skip_ws();
                v.arr.push_back(parse_value(depth + 1));
                skip_ws();
                if (i_ < n_ && p_[i_] == ',') { ++i_; continue; }
                if (i_ < n_ && p_[i_] == ']') { ++i_; return v; }
                fail("expected ',' or ']'");
            }
        }
        if (c == '"') { v.kind = JV::String; v.str = parse_string(); return v; }
        if (c == 't') { if (!lit("true")) fail("bad literal"); v.kind = JV::True; return v; }
        if (c == 'f') { if (!lit("false")) fail("bad literal"); v.kind = JV::False; return v; }
        if (c == 'n') { if (!lit("null")) fail("bad literal"); v.kind = JV::Null; return v; }
        if (c == '-' || (c >= '0' && c <= '9')) { const size_t from = i_; parse_number(); v.kind = JV::Number; for (size_t k = from; k < i_; ++k) v.str.push_back((char16_t)p_[k]); return v; }
        fail("unexpected character");
    }
    void parse_number() {
        if (p_[i_] == '-') ++i_;
        if (i_ >= n_) fail("bad number");
        if (p_[i_] == '0') ++i_;
        else if (p_[i_] >= '1' && p_[i_] <= '9') { while (i_ < n_ && p_[i_] >= '0' && p_[i_] <= '9') ++i_; }
        else fail("bad number");
        if (i_ < n_ && p_[i_] == '.') {
            ++i_;
            if (i_ >= n_ || p_[i_] < '0' || p_[i_] > '9') fail("bad number");
            while (i_ < n_ && p_[i_] >= '0' && p_[i_] <= '9') ++i_;
        }
        if (i_ < n_ && (p_[i_] == 'e' || p_[i_] == 'E')) {
            ++i_;
            if (i_ < n_ && (p_[i_] == '+' || p_[i_] == '-')) ++i_;
            if (i_ >= n_ || p_[i_] < '0' || p_[i_] > '9') fail("bad number");
            while (i_ < n_ && p_[i_] >= '0' && p_[i_] <= '9') ++i_;
        }
    }
    static void put_cp(S& out, uint32_t cp) {
        if (cp >= 0x10000) {
            cp -= 0x10000;
            out.push_back((char16_t)(0xD800 + (cp >> 10)));
            out.push_back((char16_t)(0xDC00 + (cp & 0x3FF)));
        } else {
            out.push_back((char16_t)cp);
        }
    }
    int hex4() {
        if (i_ + 4 > n_) fail("bad \\u escape");
        int v = 0;
        for (int k = 0; k < 4; ++k) {
            uint8_t h = p_[i_ + k];
            int d = (h >= '0' && h <= '9') ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10 : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : -1;
            if (d < 0) fail("bad \\u escape");
            v = v * 16 + d;
        }
        i_ += 4;
        return v;
    }
    S parse_string() {
        ++i_;  // opening quote
        S out;
        for (;;) {
            if (i_ >= n_) fail("unterminated string");
            uint8_t c = p_[i_];
            if (c == '"') { ++i_; return out; }
            if (c < 0x20) fail("control character in string");
            if (c == '\\') {
                ++i_;
                if (i_ >= n_) fail("bad escape");
                uint8_t e = p_[i_++];
                switch (e) {
                    case '"': out.push_back('"'); break;
                    case '\\': out.push_back('\\'); break;
                    case '/': out.push_back('/'); break;
                    case 'b': out.push_back('\b'); break;
                    case 'f': out.push_back('\f'); break;
                    case 'n': out.push_back('\n'); break;
                    case 'r': out.push_back('\r'); break;
                    case 't': out.push_back('\t'); break;
                    case 'u': out.push_back((char16_t)hex4()); break;
                    default: fail("bad escape");
                }
                continue;
            }
            if (c < 0x80) { out.push_back(c); ++i_; continue; }
            // UTF-8 multi-byte; malformed input is an error for System.Text.Json.
            int len = (c >= 0xF0) ? 4 : (c >= 0xE0) ? 3 : (c >= 0xC2) ? 2 : 0;
            if (!len || i_ + len > n_) fail("invalid UTF-8 in string");
            uint32_t cp = (len == 2) ? (c & 0x1Fu) : (len == 3) ? (c & 0x0Fu) : (c & 0x07u);
            for (int k = 1; k < len; ++k) {
                uint8_t cc = p_[i_ + k];
                if ((cc & 0xC0) != 0x80) fail("invalid UTF-8 in string");
                cp = (cp << 6) | (cc & 0x3Fu);
            }
            if ((len == 3 && cp < 0x800) || (len == 4 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF))
                fail("invalid UTF-8 in string");
            put_cp(out, cp);
            i_ += len;
        }
    }
};

}  // namespace orc
