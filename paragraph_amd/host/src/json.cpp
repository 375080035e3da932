// common::Json -- recursive-descent reader and writer (RFC 8259).  See include/common/Json.hh.
#include "common/Json.hh"

#include <cerrno>
#include <charconv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

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
                obj[std::move(key)] = value(depth + 1);
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
    // The text is printf's "%.{p}g" for the smallest p in 15, 16, 17 that reads back as the same double (what a jsoncpp build
    // writes), plus ".0" where that leaves a bare integer.  Made without the three snprintf / strtod rounds: the shortest digit
    // string that reads back (std::to_chars) has n <= 17 digits, the p-digit rounding for n <= p is that string padded with zeros
    // (which %g strips), so p = max(15, n) and only %g's choice of notation is left: scientific iff the decimal exponent
    // X < -4 or X >= p.  (A genotype document holds some sixty doubles; this was most of the time of writing it.)
    if (d != 0.0 && std::fabs(d) < 2.2250738585072014e-308)
    {
        // subnormal numbers carry fewer than 15 significant digits: the p-digit rounding is then NOT the shortest string padded
        // with zeros -- these (never seen in a document) take the literal route
        char slow[40];
        for (int prec = 15; prec <= 17; ++prec)
        {
            snprintf(slow, sizeof slow, "%.*g", prec, d);
            if (strtod(slow, nullptr) == d)
                break;
        }
        out += slow;
        return;
    }
    char sci[40];
    const auto conv = std::to_chars(sci, sci + sizeof sci, d, std::chars_format::scientific);
    // "[-]d[.ddd]e[+-]XX"
    const char* p = sci;
    const bool negative = *p == '-';
    if (negative)
        ++p;
    char digits[24];
    int n = 0;
    for (; p < conv.ptr && *p != 'e'; ++p)
        if (*p != '.')
            digits[n++] = *p;
    int X = 0;
    {
        ++p;  // 'e'
        const bool eneg = *p == '-';
        ++p;
        for (; p < conv.ptr; ++p)
            X = X * 10 + (*p - '0');
        if (eneg)
            X = -X;
    }
    const int P = n > 15 ? n : 15;
    char buf[48];
    char* w = buf;
    if (negative)
        *w++ = '-';
    if (X < -4 || X >= P)
    {
        *w++ = digits[0];
        if (n > 1)
        {
            *w++ = '.';
            for (int i = 1; i < n; ++i)
                *w++ = digits[i];
        }
        *w++ = 'e';
        *w++ = X < 0 ? '-' : '+';
        const int ax = X < 0 ? -X : X;
        if (ax >= 100)
            *w++ = (char)('0' + ax / 100);
        *w++ = (char)('0' + (ax / 10) % 10);
        *w++ = (char)('0' + ax % 10);
    }
    else if (X < 0)
    {
        *w++ = '0';
        *w++ = '.';
        for (int i = -1; i > X; --i)
            *w++ = '0';
        for (int i = 0; i < n; ++i)
            *w++ = digits[i];
    }
    else
    {
        for (int i = 0; i <= X; ++i)
            *w++ = i < n ? digits[i] : '0';
        if (n > X + 1)
        {
            *w++ = '.';
            for (int i = X + 1; i < n; ++i)
                *w++ = digits[i];
        }
        else
        {
            *w++ = '.';
            *w++ = '0';
        }
    }
    out.append(buf, (size_t)(w - buf));
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
    // open / fstat / read: a graph description is a few KB and the many-site workflow reads one per site -- a stream object with
    // its locale and buffer, and a second copy through a string stream, cost as much as parsing it
    const int fd = ::open(path.c_str(), O_RDONLY | O_CLOEXEC);
    if (fd < 0)
        throw std::runtime_error("Cannot open JSON file " + path);
    std::string text;
    struct stat st;
    if (::fstat(fd, &st) == 0 && st.st_size > 0)
        text.reserve((size_t)st.st_size);
    char buffer[16384];
    for (;;)
    {
        const ssize_t n = ::read(fd, buffer, sizeof(buffer));
        if (n < 0)
        {
            if (errno == EINTR)
                continue;
            ::close(fd);
            throw std::runtime_error("Cannot read JSON file " + path);
        }
        if (n == 0)
            break;
        text.append(buffer, (size_t)n);
    }
    ::close(fd);
    return parse(text);
}

Json::Json(Json const& o) : kind_(o.kind_), v_(o.v_)
{
    switch (kind_)
    {
    case STRING: v_.s = new std::string(*o.v_.s); break;
    case ARRAY: v_.elements = new Elements(*o.v_.elements); break;
    case OBJECT: v_.members = new Members(*o.v_.members); break;
    default: break;
    }
}

Json& Json::operator=(Json const& o)
{
    if (this != &o)
    {
        Json copy(o);  // o may live inside *this
        *this = std::move(copy);
    }
    return *this;
}

void Json::destroy() noexcept
{
    switch (kind_)
    {
    case STRING: delete v_.s; break;
    case ARRAY: delete v_.elements; break;
    case OBJECT: delete v_.members; break;
    default: break;
    }
    kind_ = NUL;
    v_.u = 0;
}

Json::Members const& Json::noMembers()
{
    static const Members none;
    return none;
}

Json::Elements const& Json::noElements()
{
    static const Elements none;
    return none;
}

Json::Elements& Json::elements()
{
    if (kind_ == NUL)
    {
        kind_ = ARRAY;
        v_.elements = new Elements();
    }
    if (kind_ != ARRAY)
        throw std::runtime_error("JSON value is not an array");
    return *v_.elements;
}

bool Json::asBool() const
{
    switch (kind_)
    {
    case BOOL: return v_.u != 0;
    case INT: return v_.i != 0;
    case UINT: return v_.u != 0;
    case NUL: return false;
    default: throw std::runtime_error("JSON value is not convertible to bool");
    }
}

int64_t Json::asInt64() const
{
    switch (kind_)
    {
    case INT: return v_.i;
    case UINT:
        if (v_.u > (uint64_t)INT64_MAX)
            throw std::runtime_error("JSON value out of int64 range");
        return (int64_t)v_.u;
    case REAL: return (int64_t)v_.d;
    case BOOL: return v_.u ? 1 : 0;
    case NUL: return 0;
    default: throw std::runtime_error("JSON value is not convertible to int");
    }
}

uint64_t Json::asUInt64() const
{
    switch (kind_)
    {
    case INT:
        if (v_.i < 0)
            throw std::runtime_error("JSON value out of uint64 range");
        return (uint64_t)v_.i;
    case UINT: return v_.u;
    case REAL:
        if (v_.d < 0)
            throw std::runtime_error("JSON value out of uint64 range");
        return (uint64_t)v_.d;
    case BOOL: return v_.u ? 1 : 0;
    case NUL: return 0;
    default: throw std::runtime_error("JSON value is not convertible to uint");
    }
}

double Json::asDouble() const
{
    switch (kind_)
    {
    case INT: return (double)v_.i;
    case UINT: return (double)v_.u;
    case REAL: return v_.d;
    case BOOL: return v_.u ? 1.0 : 0.0;
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
    return *v_.s;
}

Json& Json::member(std::string_view key)
{
    if (kind_ == NUL)
    {
        kind_ = OBJECT;
        v_.members = new Members();
    }
    if (kind_ != OBJECT)
        throw std::runtime_error("JSON value is not an object (member " + std::string(key) + ")");
    auto it = v_.members->lower_bound(key);
    if (it != v_.members->end() && it->first == key)
        return it->second;
    return v_.members->emplace_hint(it, std::string(key), Json())->second;
}

Json& Json::member(std::string&& key)
{
    if (kind_ == NUL)
    {
        kind_ = OBJECT;
        v_.members = new Members();
    }
    if (kind_ != OBJECT)
        throw std::runtime_error("JSON value is not an object (member " + key + ")");
    auto it = v_.members->lower_bound(std::string_view(key));
    if (it != v_.members->end() && it->first == key)
        return it->second;
    return v_.members->emplace_hint(it, std::move(key), Json())->second;
}

Json const& Json::member(std::string_view key) const
{
    static const Json null_value;
    if (kind_ == NUL)
        return null_value;
    if (kind_ != OBJECT)
        throw std::runtime_error("JSON value is not an object (member " + std::string(key) + ")");
    auto it = v_.members->find(key);
    return it == v_.members->end() ? null_value : it->second;
}

std::vector<std::string> Json::getMemberNames() const
{
    std::vector<std::string> names;
    for (auto const& kv : members())
        names.push_back(kv.first);
    return names;
}

Json& Json::append(Json v)
{
    Elements& e = elements();
    e.push_back(std::move(v));
    return e.back();
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
        if (kind_ == INT && v_.i < 0)
            return o.kind_ == INT && o.v_.i == v_.i;
        if (o.kind_ == INT && o.v_.i < 0)
            return false;
        return asUInt64() == o.asUInt64();
    }
    if (kind_ != o.kind_)
        return false;
    switch (kind_)
    {
    case NUL: return true;
    case BOOL: return v_.u == o.v_.u;
    case STRING: return *v_.s == *o.v_.s;
    case ARRAY: return *v_.elements == *o.v_.elements;
    case OBJECT: return *v_.members == *o.v_.members;
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
    case BOOL: out += v_.u ? "true" : "false"; break;
    case INT: out += std::to_string(v_.i); break;
    case UINT: out += std::to_string(v_.u); break;
    case REAL: real(out, v_.d); break;
    case STRING: quote(out, *v_.s); break;
    case ARRAY:
    {
        Elements const& elements_ = *v_.elements;
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
    }
    case OBJECT:
    {
        Members const& members_ = *v_.members;
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
}

std::string Json::dump(int indent) const
{
    std::string out;
    write(out, indent, 0);
    return out;
}
}  // namespace common
