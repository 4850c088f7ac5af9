// json.h — minimal JSON value / parser / styled writer for the .ism configuration schema
// (reference: utils/json_object.cpp:41-178 uses jsoncpp; only objects, arrays, strings, numbers, bools, null are needed).
#pragma once
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace ism3d {

class Json {
public:
    enum Type { Null, Bool, Number, String, Array, Object };
    Type type = Null;
    bool b = false;
    double num = 0;
    bool is_int = false;
    std::string str;
    std::vector<Json> arr;
    std::vector<std::pair<std::string, Json>> obj;   // insertion order kept for writing

    Json() {}
    static Json object() { Json j; j.type = Object; return j; }
    static Json array() { Json j; j.type = Array; return j; }
    static Json of(bool v) { Json j; j.type = Bool; j.b = v; return j; }
    static Json of(int v) { Json j; j.type = Number; j.num = v; j.is_int = true; return j; }
    static Json of(double v) { Json j; j.type = Number; j.num = v; return j; }
    static Json of(const std::string& v) { Json j; j.type = String; j.str = v; return j; }

    bool isObject() const { return type == Object; }
    bool isNull() const { return type == Null; }
    bool has(const std::string& k) const { return find(k) != nullptr; }
    const Json* find(const std::string& k) const {
        if (type != Object) return nullptr;
        for (auto& kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
    Json& operator[](const std::string& k) {
        if (type == Null) type = Object;
        for (auto& kv : obj) if (kv.first == k) return kv.second;
        obj.emplace_back(k, Json());
        return obj.back().second;
    }
    const Json& at(const std::string& k) const {
        const Json* p = find(k);
        if (!p) throw std::runtime_error("json: missing key " + k);
        return *p;
    }

    static Json parse(const std::string& text) {
        size_t i = 0;
        Json v = parseValue(text, i);
        skip(text, i);
        if (i != text.size()) throw std::runtime_error("json: trailing characters");
        return v;
    }

    std::string dump(int indent = 0) const {
        std::ostringstream os;
        write(os, indent, 0);
        return os.str();
    }

private:
    static void skip(const std::string& s, size_t& i) { while (i < s.size() && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r')) ++i; }
    static Json parseValue(const std::string& s, size_t& i) {
        skip(s, i);
        if (i >= s.size()) throw std::runtime_error("json: unexpected end");
        const char c = s[i];
        if (c == '{') {
            Json j = object(); ++i; skip(s, i);
            if (i < s.size() && s[i] == '}') { ++i; return j; }
            for (;;) {
                skip(s, i);
                if (i >= s.size() || s[i] != '"') throw std::runtime_error("json: expected key");
                std::string k = parseString(s, i);
                skip(s, i);
                if (i >= s.size() || s[i] != ':') throw std::runtime_error("json: expected ':'");
                ++i;
                Json v = parseValue(s, i);
                j.obj.emplace_back(k, v);
                skip(s, i);
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == '}') { ++i; break; }
                throw std::runtime_error("json: expected ',' or '}'");
            }
            return j;
        }
        if (c == '[') {
            Json j = array(); ++i; skip(s, i);
            if (i < s.size() && s[i] == ']') { ++i; return j; }
            for (;;) {
                j.arr.push_back(parseValue(s, i));
                skip(s, i);
                if (i < s.size() && s[i] == ',') { ++i; continue; }
                if (i < s.size() && s[i] == ']') { ++i; break; }
                throw std::runtime_error("json: expected ',' or ']'");
            }
            return j;
        }
        if (c == '"') return of(parseString(s, i));
        if (s.compare(i, 4, "true") == 0) { i += 4; return of(true); }
        if (s.compare(i, 5, "false") == 0) { i += 5; return of(false); }
        if (s.compare(i, 4, "null") == 0) { i += 4; return Json(); }
        // number
        size_t j = i;
        bool integral = true;
        while (j < s.size() && (isdigit((unsigned char)s[j]) || s[j] == '-' || s[j] == '+' || s[j] == '.' || s[j] == 'e' || s[j] == 'E')) {
            if (s[j] == '.' || s[j] == 'e' || s[j] == 'E') integral = false;
            ++j;
        }
        if (j == i) throw std::runtime_error(std::string("json: unexpected character '") + c + "'");
        Json v; v.type = Number; v.num = std::strtod(s.substr(i, j - i).c_str(), nullptr); v.is_int = integral;
        i = j;
        return v;
    }
    static std::string parseString(const std::string& s, size_t& i) {
        std::string out; ++i;
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < s.size()) {
                ++i;
                switch (s[i]) {
                    case 'n': out += '\n'; break; case 't': out += '\t'; break; case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break; case 'f': out += '\f'; break;
                    case 'u': i += 4; out += '?'; break;
                    default: out += s[i];
                }
            } else out += s[i];
            ++i;
        }
        if (i >= s.size()) throw std::runtime_error("json: unterminated string");
        ++i;
        return out;
    }
    void write(std::ostream& os, int indent, int depth) const {
        const std::string pad(indent * (depth + 1), ' '), padc(indent * depth, ' ');
        const char* nl = indent ? "\n" : "";
        switch (type) {
            case Null: os << "null"; break;
            case Bool: os << (b ? "true" : "false"); break;
            case Number:
                if (is_int) os << (long long)num;
                else { char buf[64]; snprintf(buf, sizeof buf, "%.17g", num); os << buf; }
                break;
            case String: os << '"'; for (char c : str) { if (c == '"' || c == '\\') os << '\\'; os << c; } os << '"'; break;
            case Array:
                os << '[' << nl;
                for (size_t k = 0; k < arr.size(); ++k) { os << pad; arr[k].write(os, indent, depth + 1); os << (k + 1 < arr.size() ? "," : "") << nl; }
                os << padc << ']';
                break;
            case Object:
                os << '{' << nl;
                for (size_t k = 0; k < obj.size(); ++k) {
                    os << pad << '"' << obj[k].first << "\" : ";
                    obj[k].second.write(os, indent, depth + 1);
                    os << (k + 1 < obj.size() ? "," : "") << nl;
                }
                os << padc << '}';
                break;
        }
    }
};

}  // namespace ism3d
