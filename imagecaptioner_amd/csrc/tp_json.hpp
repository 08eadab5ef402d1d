// JSON reader for the table loaders (lexicon.json, actions.json,
// required_facets.json, profile, JSON vocabularies).
//
// Nodes live in one flat vector; object members and array items are sibling
// chains in document order.  The behaviours that matter are those of
// System.Text.Json as the reference uses it (JsonDocument.Parse with default
// options, Program.cs:1018, 1277, 1309, 1376): strict grammar, duplicate
// property names kept, lookup by name returns the last duplicate, depth 64.
#pragma once
#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

namespace tp {

struct JsonFail : std::runtime_error {
    explicit JsonFail(const std::string& m) : std::runtime_error(m) {}
};

enum class JKind : uint8_t { Null, Bool, Number, String, Array, Object };

struct JNode {
    JKind kind = JKind::Null;
    int32_t first = -1;   // first child (Array: item, Object: member)
    int32_t next = -1;    // next sibling
    int32_t name = -1;    // index into JDoc::text of the member name (Object children only)
    int32_t value = -1;   // index into JDoc::text for String, and for Number (its characters as written)
};

class JDoc {
public:
    std::vector<JNode> nodes;
    std::vector<std::u16string> text;
    int32_t root = -1;

    static JDoc parse(const uint8_t* data, size_t len) {
        JDoc d;
        Cursor c{data, len, 0};
        if (len >= 3 && data[0] == 0xEF && data[1] == 0xBB && data[2] == 0xBF) c.pos = 3;
        c.blanks();
        d.root = d.value(c, 0);
        c.blanks();
        if (c.pos != c.len) c.die("unexpected data after the document");
        return d;
    }

    JKind kind(int32_t n) const { return nodes[n].kind; }
    const std::u16string& str(int32_t n) const { return text[nodes[n].value]; }
    const std::u16string& member_name(int32_t n) const { return text[nodes[n].name]; }
    // JsonElement.TryGetProperty: throws unless the element is an object; the last duplicate wins.
    int32_t member(int32_t obj, const char16_t* name) const {
        if (nodes[obj].kind != JKind::Object) throw JsonFail("property lookup on a JSON value that is not an object");
        int32_t hit = -1;
        for (int32_t c = nodes[obj].first; c >= 0; c = nodes[c].next)
            if (text[nodes[c].name] == name) hit = c;
        return hit;
    }
    // GetString() ?? "": String gives its text, Null gives "", anything else throws.
    std::u16string string_or_empty(int32_t n) const {
        if (nodes[n].kind == JKind::String) return str(n);
        if (nodes[n].kind == JKind::Null) return std::u16string();
        throw JsonFail("a JSON string was expected");
    }

private:
    struct Cursor {
        const uint8_t* p;
        size_t len, pos;
        [[noreturn]] void die(const char* what) const { throw JsonFail(std::string("JSON: ") + what + " (offset " + std::to_string(pos) + ")"); }
        bool more() const { return pos < len; }
        uint8_t peek() const { return p[pos]; }
        void blanks() { while (pos < len && (p[pos] == 0x20 || p[pos] == 0x09 || p[pos] == 0x0A || p[pos] == 0x0D)) ++pos; }
        void expect_word(const char* w) {
            for (; *w; ++w, ++pos) if (pos >= len || p[pos] != (uint8_t)*w) die("bad literal");
        }
    };

    int32_t add(JKind k) { nodes.emplace_back(); nodes.back().kind = k; return (int32_t)nodes.size() - 1; }
    int32_t add_text(std::u16string s) { text.push_back(std::move(s)); return (int32_t)text.size() - 1; }

    int32_t value(Cursor& c, int depth) {
        if (depth > 64) c.die("nesting deeper than 64");
        if (!c.more()) c.die("document ends early");
        switch (c.peek()) {
            case '{': return object(c, depth);
            case '[': return array(c, depth);
            case '"': { int32_t n = add(JKind::String); int32_t tix = add_text(string(c)); nodes[n].value = tix; return n; }
            case 't': c.expect_word("true"); return add(JKind::Bool);
            case 'f': c.expect_word("false"); return add(JKind::Bool);
            case 'n': c.expect_word("null"); return add(JKind::Null);
            default: { const size_t from = c.pos; number(c); int32_t n = add(JKind::Number); std::u16string lit; for (size_t k = from; k < c.pos; ++k) lit.push_back((char16_t)c.p[k]); int32_t tix = add_text(std::move(lit)); nodes[n].value = tix; return n; }
        }
    }
    int32_t object(Cursor& c, int depth) {
        int32_t self = add(JKind::Object);
        ++c.pos;
        c.blanks();
        if (c.more() && c.peek() == '}') { ++c.pos; return self; }
        int32_t tail = -1;
        while (true) {
            c.blanks();
            if (!c.more() || c.peek() != '"') c.die("a property name was expected");
            int32_t nm = add_text(string(c));
            c.blanks();
            if (!c.more() || c.peek() != ':') c.die("':' was expected");
            ++c.pos;
            c.blanks();
            int32_t child = value(c, depth + 1);
            nodes[child].name = nm;
            if (tail < 0) nodes[self].first = child; else nodes[tail].next = child;
            tail = child;
            c.blanks();
            if (!c.more()) c.die("object not closed");
            if (c.peek() == ',') { ++c.pos; continue; }
            if (c.peek() == '}') { ++c.pos; return self; }
            c.die("',' or '}' was expected");
        }
    }
    int32_t array(Cursor& c, int depth) {
        int32_t self = add(JKind::Array);
        ++c.pos;
        c.blanks();
        if (c.more() && c.peek() == ']') { ++c.pos; return self; }
        int32_t tail = -1;
        while (true) {
            c.blanks();
            int32_t child = value(c, depth + 1);
            if (tail < 0) nodes[self].first = child; else nodes[tail].next = child;
            tail = child;
            c.blanks();
            if (!c.more()) c.die("array not closed");
            if (c.peek() == ',') { ++c.pos; continue; }
            if (c.peek() == ']') { ++c.pos; return self; }
            c.die("',' or ']' was expected");
        }
    }
    static void number(Cursor& c) {
        auto digit = [&]() { return c.more() && c.peek() >= '0' && c.peek() <= '9'; };
        if (c.more() && c.peek() == '-') ++c.pos;
        if (!digit()) c.die("a value was expected");
        if (c.peek() == '0') ++c.pos; else while (digit()) ++c.pos;
        if (c.more() && c.peek() == '.') { ++c.pos; if (!digit()) c.die("bad number"); while (digit()) ++c.pos; }
        if (c.more() && (c.peek() == 'e' || c.peek() == 'E')) {
            ++c.pos;
            if (c.more() && (c.peek() == '+' || c.peek() == '-')) ++c.pos;
            if (!digit()) c.die("bad number");
            while (digit()) ++c.pos;
        }
    }
    static std::u16string string(Cursor& c) {
        std::u16string out;
        ++c.pos;
        while (true) {
            if (!c.more()) c.die("string not closed");
            uint32_t b = c.p[c.pos];
            if (b == '"') { ++c.pos; return out; }
            if (b < 0x20) c.die("raw control character in a string");
            if (b == '\\') {
                if (c.pos + 1 >= c.len) c.die("bad escape");
                uint8_t e = c.p[c.pos + 1];
                c.pos += 2;
                if (e == 'u') {
                    if (c.pos + 4 > c.len) c.die("bad \\u escape");
                    uint32_t v = 0;
                    for (int k = 0; k < 4; ++k) {
                        uint8_t h = c.p[c.pos + k];
                        uint32_t d = (h >= '0' && h <= '9') ? h - '0' : (h >= 'a' && h <= 'f') ? h - 'a' + 10 : (h >= 'A' && h <= 'F') ? h - 'A' + 10 : 99;
                        if (d == 99) c.die("bad \\u escape");
                        v = (v << 4) | d;
                    }
                    c.pos += 4;
                    out.push_back((char16_t)v);
                } else {
                    const char* from = "\"\\/bfnrt";
                    const char to[] = {'"', '\\', '/', '\b', '\f', '\n', '\r', '\t'};
                    int k = 0;
                    while (from[k] && from[k] != (char)e) ++k;
                    if (!from[k]) c.die("bad escape");
                    out.push_back((char16_t)to[k]);
                }
                continue;
            }
            if (b < 0x80) { out.push_back((char16_t)b); ++c.pos; continue; }
            // multi-byte UTF-8, validated
            int extra = (b >= 0xF0 && b <= 0xF4) ? 3 : (b >= 0xE0) ? 2 : (b >= 0xC2) ? 1 : -1;
            if (extra < 0 || c.pos + (size_t)extra >= c.len) c.die("malformed UTF-8");
            uint32_t cp = extra == 1 ? (b & 0x1F) : extra == 2 ? (b & 0x0F) : (b & 0x07);
            for (int k = 1; k <= extra; ++k) {
                uint8_t cc = c.p[c.pos + k];
                if ((cc & 0xC0) != 0x80) c.die("malformed UTF-8");
                cp = (cp << 6) | (cc & 0x3F);
            }
            if ((extra == 2 && cp < 0x800) || (extra == 3 && (cp < 0x10000 || cp > 0x10FFFF)) || (cp >= 0xD800 && cp <= 0xDFFF)) c.die("malformed UTF-8");
            if (cp >= 0x10000) { cp -= 0x10000; out.push_back((char16_t)(0xD800 | (cp >> 10))); out.push_back((char16_t)(0xDC00 | (cp & 0x3FF))); }
            else out.push_back((char16_t)cp);
            c.pos += extra + 1;
        }
    }
};

}  // namespace tp
