// common::Json -- recursive-descent reader and writer (RFC 8259).  See include/common/Json.hh.
#include "common/Json.hh"

#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace common
{
namespace
{
struct Reader
{
    const char* p;
    const char* end;
    const char* begin;

    [[noreturn]] void fail(const char* what) const
    {
        size_t line = 1, col = 1;
        for (const char* q = begin; q < p; ++q)
        {
            if (*q == '\n')
            {
                ++line;
                col = 1;
            }
            else
                ++col;
        }
        throw std::runtime_error("JSON: " + std::string(what) + " at " + std::to_string(line) + ":" + std::to_string(col));
    }
    void skipSpace()
    {
        while (p < end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r'))
            ++p;
    }
    bool take(char c)
    {
        skipSpace();
        if (p < end && *p == c)
        {
            ++p;
            return true;
        }
        return false;
    }
    void word(const char* w)
    {
        size_t n = strlen(w);
        if ((size_t)(end - p) < n || memcmp(p, w, n) != 0)
            fail("unexpected token");
        p += n;
    }
    static void utf8(std::string& out, unsigned cp)
    {
        if (cp < 0x80)
            out += (char)cp;
        else if (cp < 0x800)
        {
            out += (char)(0xC0 | (cp >> 6));
            out += (char)(0x80 | (cp & 0x3F));
        }
        else if (cp < 0x10000)
        {
            out += (char)(0xE0 | (cp >> 12));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
        else
        {
            out += (char)(0xF0 | (cp >> 18));
            out += (char)(0x80 | ((cp >> 12) & 0x3F));
            out += (char)(0x80 | ((cp >> 6) & 0x3F));
            out += (char)(0x80 | (cp & 0x3F));
        }
    }
    unsigned hex4()
    {
        if (end - p < 4)
            fail("short \\u escape");
        unsigned v = 0;
        for (int i = 0; i < 4; ++i, ++p)
        {
            char c = *p;
            v <<= 4;
            if (c >= '0' && c <= '9')
                v |= (unsigned)(c - '0');
            else if (c >= 'a' && c <= 'f')
                v |= (unsigned)(c - 'a' + 10);
            else if (c >= 'A' && c <= 'F')
                v |= (unsigned)(c - 'A' + 10);
            else
                fail("bad \\u escape");
        }
        return v;
    }
    std::string string()
    {
        std::string out;
        ++p;  // opening quote
        for (;;)
        {
            if (p >= end)
                fail("unterminated string");
            char c = *p++;
            if (c == '"')
                return out;
            if (c != '\\')
            {
                out += c;
                continue;
            }
            if (p >= end)
                fail("unterminated escape");
            char e = *p++;
            switch (e)
            {
            case '"': out += '"'; break;
            case '\\': out += '\\'; break;
            case '/': out += '/'; break;
            case 'b': out += '\b'; break;
            case 'f': out += '\f'; break;
            case 'n': out += '\n'; break;
            case 'r': out += '\r'; break;
            case 't': out += '\t'; break;
            case 'u':
            {
                unsigned cp = hex4();
                if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u')
                {
                    p += 2;
                    unsigned lo = hex4();
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                utf8(out, cp);
                break;
            }
            default: fail("unknown escape");
            }
        }
    }
    Json number()
    {
        const char* s = p;
        bool real = false;
        if (p < end && *p == '-')
            ++p;
        while (p < end && ((*p >= '0' && *p <= '9') || *p == '.' || *p == 'e' || *p == 'E' || *p == '+' || *p == '-'))
        {
            if (*p == '.' || *p == 'e' || *p == 'E')
                real = true;
            ++p;
        }
        std::string tok(s, p);
        if (tok.empty() || tok == "-")
            fail("bad number");
        if (!real)
        {
            errno = 0;
            char* e = nullptr;
            if (tok[0] == '-')
            {
                long long v = strtoll(tok.c_str(), &e, 10);
                if (errno == 0 && *e == 0)
                    return Json((int64_t)v);
            }
            else
            {
                unsigned long long v = strtoull(tok.c_str(), &e, 10);
                if (errno == 0 && *e == 0)
                    return v <= (unsigned long long)INT64_MAX ? Json((int64_t)v) : Json((uint64_t)v);
            }
        }
        char* e = nullptr;
        double d = strtod(tok.c_str(), &e);
        if (*e != 0)
            fail("bad number");
        return Json(d);
    }
    Json value(int depth)
    {
        if (depth > 512)
            fail("nesting too deep");
        skipSpace();
        if (p >= end)
            fail("unexpected end of input");
        switch (*p)
        {
        case '{':
        {
            ++p;
            Json obj = Json::object();
            if (take('}'))
                return obj;
            for (;;)
            {
                skipSpace();
                if (p >= end || *p != '"')
                    fail("expected member name");
                std::string key = string();
                if (!take(':'))
                    fail("expected ':'");
                obj[key] = value(depth + 1);
                if (take(','))
                    continue;
                if (take('}'))
                    return obj;
                fail("expected ',' or '}'");
            }
        }
        case '[':
        {
            ++p;
            Json arr = Json::array();
            if (take(']'))
                return arr;
            for (;;)
            {
                arr.append(value(depth + 1));
                if (take(','))
                    continue;
                if (take(']'))
                    return arr;
                fail("expected ',' or ']'");
            }
        }
        case '"': return Json(string());
        case 't': word("true"); return Json(true);
        case 'f': word("false"); return Json(false);
        case 'n': word("null"); return Json();
        default: return number();
        }
    }
};

void quote(std::string& out, std::string const& s)
{
    out += '"';
    for (unsigned char c : s)
    {
        switch (c)
        {
        case '"': out += "\\\""; break;
        case '\\': out += "\\\\"; break;
        case '\b': out += "\\b"; break;
        case '\f': out += "\\f"; break;
        case '\n': out += "\\n"; break;
        case '\r': out += "\\r"; break;
        case '\t': out += "\\t"; break;
        default:
            if (c < 0x20)
            {
                char buf[8];
                snprintf(buf, sizeof buf, "\\u%04x", c);
                out += buf;
            }
            else
                out += (char)c;
        }
    }
    out += '"';
}

void real(std::string& out, double d)
{
    if (std::isnan(d) || std::isinf(d))
    {
        out += "null";  // what jsoncpp's writer emits without useSpecialFloats
        return;
    }
    char buf[40];
    for (int prec = 15; prec <= 17; ++prec)
    {
        snprintf(buf, sizeof buf, "%.*g", prec, d);
        if (strtod(buf, nullptr) == d)
            break;
    }
    out += buf;
    if (!strpbrk(buf, ".eEn"))
        out += ".0";
}
}  // namespace

Json Json::parse(std::string const& text)
{
    Reader r{ text.data(), text.data() + text.size(), text.data() };
    Json v = r.value(0);
    r.skipSpace();
    if (r.p != r.end)
        r.fail("trailing characters");
    return v;
}

Json Json::parseFile(std::string const& path)
{
    std::ifstream in(path, std::ios::binary);
    if (!in.good())
        throw std::runtime_error("Cannot open JSON file " + path);
    std::stringstream ss;
    ss << in.rdbuf();
    return parse(ss.str());
}

bool Json::asBool() const
{
    switch (kind_)
    {
    case BOOL: return u_ != 0;
    case INT: return i_ != 0;
    case UINT: return u_ != 0;
    case NUL: return false;
    default: throw std::runtime_error("JSON value is not convertible to bool");
    }
}

int64_t Json::asInt64() const
{
    switch (kind_)
    {
    case INT: return i_;
    case UINT:
        if (u_ > (uint64_t)INT64_MAX)
            throw std::runtime_error("JSON value out of int64 range");
        return (int64_t)u_;
    case REAL: return (int64_t)d_;
    case BOOL: return u_ ? 1 : 0;
    case NUL: return 0;
    default: throw std::runtime_error("JSON value is not convertible to int");
    }
}

uint64_t Json::asUInt64() const
{
    switch (kind_)
    {
    case INT:
        if (i_ < 0)
            throw std::runtime_error("JSON value out of uint64 range");
        return (uint64_t)i_;
    case UINT: return u_;
    case REAL:
        if (d_ < 0)
            throw std::runtime_error("JSON value out of uint64 range");
        return (uint64_t)d_;
    case BOOL: return u_ ? 1 : 0;
    case NUL: return 0;
    default: throw std::runtime_error("JSON value is not convertible to uint");
    }
}

double Json::asDouble() const
{
    switch (kind_)
    {
    case INT: return (double)i_;
    case UINT: return (double)u_;
    case REAL: return d_;
    case BOOL: return u_ ? 1.0 : 0.0;
    case NUL: return 0.0;
    default: throw std::runtime_error("JSON value is not convertible to double");
    }
}

std::string const& Json::asString() const
{
    static const std::string empty;
    if (kind_ == NUL)
        return empty;
    if (kind_ != STRING)
        throw std::runtime_error("JSON value is not a string");
    return s_;
}

Json& Json::operator[](std::string const& key)
{
    if (kind_ == NUL)
        kind_ = OBJECT;
    if (kind_ != OBJECT)
        throw std::runtime_error("JSON value is not an object (member " + key + ")");
    return members_[key];
}

Json const& Json::operator[](std::string const& key) const
{
    static const Json null_value;
    if (kind_ == NUL)
        return null_value;
    if (kind_ != OBJECT)
        throw std::runtime_error("JSON value is not an object (member " + key + ")");
    auto it = members_.find(key);
    return it == members_.end() ? null_value : it->second;
}

std::vector<std::string> Json::getMemberNames() const
{
    std::vector<std::string> names;
    for (auto const& kv : members_)
        names.push_back(kv.first);
    return names;
}

Json& Json::append(Json v)
{
    if (kind_ == NUL)
        kind_ = ARRAY;
    if (kind_ != ARRAY)
        throw std::runtime_error("JSON value is not an array");
    elements_.push_back(std::move(v));
    return elements_.back();
}

bool Json::operator==(Json const& o) const
{
    if (isNumber() && o.isNumber())
    {
        if (kind_ == REAL || o.kind_ == REAL)
        {
            const double a = asDouble(), b = o.asDouble();
            return a == b || (std::isnan(a) && std::isnan(b));  // both are written as null: the same document
        }
        if (kind_ == INT && i_ < 0)
            return o.kind_ == INT && o.i_ == i_;
        if (o.kind_ == INT && o.i_ < 0)
            return false;
        return asUInt64() == o.asUInt64();
    }
    if (kind_ != o.kind_)
        return false;
    switch (kind_)
    {
    case NUL: return true;
    case BOOL: return u_ == o.u_;
    case STRING: return s_ == o.s_;
    case ARRAY: return elements_ == o.elements_;
    case OBJECT: return members_ == o.members_;
    default: return false;
    }
}

void Json::write(std::string& out, int indent, int depth) const
{
    auto newline = [&](int d) {
        if (indent >= 0)
        {
            out += '\n';
            out.append((size_t)(indent * d), ' ');
        }
    };
    switch (kind_)
    {
    case NUL: out += "null"; break;
    case BOOL: out += u_ ? "true" : "false"; break;
    case INT: out += std::to_string(i_); break;
    case UINT: out += std::to_string(u_); break;
    case REAL: real(out, d_); break;
    case STRING: quote(out, s_); break;
    case ARRAY:
        if (elements_.empty())
        {
            out += "[]";
            break;
        }
        out += '[';
        for (size_t i = 0; i < elements_.size(); ++i)
        {
            if (i)
                out += ',';
            newline(depth + 1);
            elements_[i].write(out, indent, depth + 1);
        }
        newline(depth);
        out += ']';
        break;
    case OBJECT:
        if (members_.empty())
        {
            out += "{}";
            break;
        }
        out += '{';
        {
            bool first = true;
            for (auto const& kv : members_)
            {
                if (!first)
                    out += ',';
                first = false;
                newline(depth + 1);
                quote(out, kv.first);
                out += indent >= 0 ? ": " : ":";
                kv.second.write(out, indent, depth + 1);
            }
        }
        newline(depth);
        out += '}';
        break;
    }
}

std::string Json::dump(int indent) const
{
    std::string out;
    write(out, indent, 0);
    return out;
}
}  // namespace common
