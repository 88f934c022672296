// json_min.hpp -- small strict JSON reader for the device / filter files.
//
// The reference parses these with libjansson (json_loadf with
// JSON_REJECT_DUPLICATES, src/fir.c:87, src/device.c:594).  What the loaders
// rely on, and what this reader therefore reproduces:
//   * RFC 8259 syntax, UTF-8 text, duplicate object keys are an error;
//   * a number is an INTEGER when its text has no '.', 'e' or 'E', otherwise
//     a REAL (json_is_integer / json_is_number distinguish them,
//     src/device.c:94-115, src/fir.c:143, :217); reals go through strtod;
//   * object member order is irrelevant, array order is kept.
#pragma once

#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace ookd {
namespace json {

enum class Kind { Null, Bool, Integer, Real, String, Array, Object };

struct Value {
    Kind kind = Kind::Null;
    bool b = false;
    long long i = 0;
    double d = 0.0;
    std::string s;
    std::vector<Value> arr;
    std::vector<std::pair<std::string, Value>> obj;

    bool is_object() const { return kind == Kind::Object; }
    bool is_array() const { return kind == Kind::Array; }
    bool is_string() const { return kind == Kind::String; }
    bool is_integer() const { return kind == Kind::Integer; }
    bool is_number() const { return kind == Kind::Integer || kind == Kind::Real; }
    // json_number_value: integer or real as double
    double number() const { return kind == Kind::Integer ? (double)i : d; }

    // json_object_get: nullptr when absent or when *this is not an object
    const Value *get(const char *key) const {
        if (kind != Kind::Object) return nullptr;
        for (const auto &kv : obj) {
            if (kv.first == key) return &kv.second;
        }
        return nullptr;
    }
};

struct ParseError {
    int line = 0;
    int column = 0;
    std::string text;
};

class Parser {
  public:
    Parser(const char *data, size_t len) : p_(data), end_(data + len) {}

    bool parse(Value &out, ParseError &err) {
        skip_ws();
        if (!value(out, 0)) {
            err = err_;
            return false;
        }
        skip_ws();
        if (p_ != end_) {
            fail("end of file expected");
            err = err_;
            return false;
        }
        return true;
    }

  private:
    const char *p_;
    const char *end_;
    int line_ = 1;
    int col_ = 0;
    ParseError err_;

    static const int kMaxDepth = 2048;

    bool fail(const char *msg) {
        if (err_.text.empty()) {
            err_.line = line_;
            err_.column = col_;
            err_.text = msg;
        }
        return false;
    }

    int peek() const { return p_ < end_ ? (unsigned char)*p_ : -1; }

    int next() {
        if (p_ >= end_) return -1;
        int c = (unsigned char)*p_++;
        if (c == '\n') {
            line_++;
            col_ = 0;
        } else {
            col_++;
        }
        return c;
    }

    void skip_ws() {
        for (;;) {
            int c = peek();
            if (c == ' ' || c == '\t' || c == '\n' || c == '\r') {
                next();
            } else {
                return;
            }
        }
    }

    bool literal(const char *word) {
        for (const char *w = word; *w; w++) {
            if (next() != (unsigned char)*w) return fail("invalid token");
        }
        return true;
    }

    static void append_utf8(std::string &s, uint32_t cp) {
        if (cp < 0x80) {
            s.push_back((char)cp);
        } else if (cp < 0x800) {
            s.push_back((char)(0xC0 | (cp >> 6)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        } else if (cp < 0x10000) {
            s.push_back((char)(0xE0 | (cp >> 12)));
            s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        } else {
            s.push_back((char)(0xF0 | (cp >> 18)));
            s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
            s.push_back((char)(0x80 | ((cp >> 6) & 0x3F)));
            s.push_back((char)(0x80 | (cp & 0x3F)));
        }
    }

    bool hex4(uint32_t &out) {
        out = 0;
        for (int k = 0; k < 4; k++) {
            int c = next();
            uint32_t v;
            if (c >= '0' && c <= '9') v = c - '0';
            else if (c >= 'a' && c <= 'f') v = 10 + c - 'a';
            else if (c >= 'A' && c <= 'F') v = 10 + c - 'A';
            else return fail("invalid escape");
            out = (out << 4) | v;
        }
        return true;
    }

    bool string(std::string &out) {
        if (next() != '"') return fail("string expected");
        out.clear();
        for (;;) {
            int c = next();
            if (c < 0) return fail("premature end of input");
            if (c == '"') return true;
            if (c < 0x20) return fail("control character in string");
            if (c != '\\') {
                out.push_back((char)c);
                continue;
            }
            c = next();
            switch (c) {
            case '"': out.push_back('"'); break;
            case '\\': out.push_back('\\'); break;
            case '/': out.push_back('/'); break;
            case 'b': out.push_back('\b'); break;
            case 'f': out.push_back('\f'); break;
            case 'n': out.push_back('\n'); break;
            case 'r': out.push_back('\r'); break;
            case 't': out.push_back('\t'); break;
            case 'u': {
                uint32_t cp;
                if (!hex4(cp)) return false;
                if (cp >= 0xD800 && cp <= 0xDBFF) {
                    if (next() != '\\' || next() != 'u') return fail("invalid Unicode");
                    uint32_t lo;
                    if (!hex4(lo)) return false;
                    if (lo < 0xDC00 || lo > 0xDFFF) return fail("invalid Unicode");
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                    return fail("invalid Unicode");
                }
                if (cp == 0) return fail("\\u0000 is not allowed");
                append_utf8(out, cp);
                break;
            }
            default:
                return fail("invalid escape");
            }
        }
    }

    bool number(Value &out) {
        const char *start = p_;
        bool real = false;
        if (peek() == '-') next();
        int c = peek();
        if (c == '0') {
            next();
            if (peek() >= '0' && peek() <= '9') return fail("invalid token");
        } else if (c >= '1' && c <= '9') {
            while (peek() >= '0' && peek() <= '9') next();
        } else {
            return fail("invalid token");
        }
        if (peek() == '.') {
            real = true;
            next();
            if (!(peek() >= '0' && peek() <= '9')) return fail("invalid token");
            while (peek() >= '0' && peek() <= '9') next();
        }
        if (peek() == 'e' || peek() == 'E') {
            real = true;
            next();
            if (peek() == '+' || peek() == '-') next();
            if (!(peek() >= '0' && peek() <= '9')) return fail("invalid token");
            while (peek() >= '0' && peek() <= '9') next();
        }
        std::string text(start, p_);
        errno = 0;
        if (real) {
            char *endp = nullptr;
            double v = strtod(text.c_str(), &endp);
            if (errno == ERANGE && std::isinf(v)) return fail("real number overflow");
            out.kind = Kind::Real;
            out.d = v;
        } else {
            char *endp = nullptr;
            long long v = strtoll(text.c_str(), &endp, 10);
            if (errno == ERANGE) return fail("too big integer");
            out.kind = Kind::Integer;
            out.i = v;
        }
        return true;
    }

    bool value(Value &out, int depth) {
        if (depth > kMaxDepth) return fail("maximum parsing depth reached");
        int c = peek();
        if (c == '{') {
            next();
            out.kind = Kind::Object;
            skip_ws();
            if (peek() == '}') {
                next();
                return true;
            }
            for (;;) {
                skip_ws();
                std::string key;
                if (peek() != '"') return fail("string or '}' expected");
                if (!string(key)) return false;
                for (const auto &kv : out.obj) {
                    if (kv.first == key) return fail("duplicate object key");
                }
                skip_ws();
                if (next() != ':') return fail("':' expected");
                skip_ws();
                out.obj.emplace_back(std::move(key), Value());
                if (!value(out.obj.back().second, depth + 1)) return false;
                skip_ws();
                int d = next();
                if (d == '}') return true;
                if (d != ',') return fail("'}' expected");
            }
        }
        if (c == '[') {
            next();
            out.kind = Kind::Array;
            skip_ws();
            if (peek() == ']') {
                next();
                return true;
            }
            for (;;) {
                skip_ws();
                out.arr.emplace_back();
                if (!value(out.arr.back(), depth + 1)) return false;
                skip_ws();
                int d = next();
                if (d == ']') return true;
                if (d != ',') return fail("']' expected");
            }
        }
        if (c == '"') {
            out.kind = Kind::String;
            return string(out.s);
        }
        if (c == 't') {
            out.kind = Kind::Bool;
            out.b = true;
            return literal("true");
        }
        if (c == 'f') {
            out.kind = Kind::Bool;
            out.b = false;
            return literal("false");
        }
        if (c == 'n') {
            out.kind = Kind::Null;
            return literal("null");
        }
        if (c == '-' || (c >= '0' && c <= '9')) return number(out);
        if (c < 0) return fail("premature end of input");
        return fail("invalid token");
    }
};

}  // namespace json
}  // namespace ookd
