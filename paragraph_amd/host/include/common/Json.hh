// Small JSON document model for the host workflow: graph descriptions in (share/schema/graph_schema.json), count / genotype
// documents out (share/schema/output_schema.json).  Stands where the reference uses jsoncpp's Json::Value; objects keep
// their keys sorted like Json::Value does, so documents written from here list members in the same order.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <string_view>
#include <vector>

namespace common
{
class Json
{
public:
    enum Kind { NUL, BOOL, INT, UINT, REAL, STRING, ARRAY, OBJECT };
    typedef std::map<std::string, Json, std::less<>> Members;  // (transparent comparison: looked up by string_view, no temporary)
    typedef std::vector<Json> Elements;

    // A value is 16 bytes: its kind and either the scalar itself or a pointer to its string / elements / members (a count
    // document holds a few hundred values; they are built, read back by the genotyper and dropped for every site and sample).
    Json() : kind_(NUL) { v_.u = 0; }
    Json(bool v) : kind_(BOOL) { v_.u = v; }
    Json(int v) : kind_(INT) { v_.i = v; }
    Json(int64_t v) : kind_(INT) { v_.i = v; }
    Json(unsigned v) : kind_(UINT) { v_.u = v; }
    Json(uint64_t v) : kind_(UINT) { v_.u = v; }
    Json(double v) : kind_(REAL) { v_.d = v; }
    Json(const char* v) : kind_(STRING) { v_.s = new std::string(v); }
    Json(std::string v) : kind_(STRING) { v_.s = new std::string(std::move(v)); }
    Json(Json const& o);
    Json(Json&& o) noexcept : kind_(o.kind_), v_(o.v_)
    {
        o.kind_ = NUL;
        o.v_.u = 0;
    }
    Json& operator=(Json const& o);
    Json& operator=(Json&& o) noexcept
    {
        if (this != &o)
        {
            destroy();
            kind_ = o.kind_;
            v_ = o.v_;
            o.kind_ = NUL;
            o.v_.u = 0;
        }
        return *this;
    }
    ~Json() { destroy(); }
    static Json array()
    {
        Json j;
        j.kind_ = ARRAY;
        j.v_.elements = new Elements();
        return j;
    }
    static Json object()
    {
        Json j;
        j.kind_ = OBJECT;
        j.v_.members = new Members();
        return j;
    }
    static Json parse(std::string const& text);       // throws std::runtime_error with line:column
    static Json parseFile(std::string const& path);   // throws if the file cannot be read
    std::string dump(int indent = -1) const;           // indent < 0: one line; NaN / inf are written as null

    Kind kind() const { return kind_; }
    bool isNull() const { return kind_ == NUL; }
    bool isBool() const { return kind_ == BOOL; }
    bool isNumber() const { return kind_ == INT || kind_ == UINT || kind_ == REAL; }
    bool isString() const { return kind_ == STRING; }
    bool isArray() const { return kind_ == ARRAY; }
    bool isObject() const { return kind_ == OBJECT; }

    // conversions throw std::runtime_error when the value has another kind
    bool asBool() const;
    int64_t asInt64() const;
    uint64_t asUInt64() const;
    double asDouble() const;
    std::string const& asString() const;

    // objects; a null value silently becomes an object (array) on first write, like Json::Value
    bool isMember(std::string_view key) const { return kind_ == OBJECT && v_.members->find(key) != v_.members->end(); }
    bool isMember(std::string const& key) const { return isMember(std::string_view(key)); }
    bool isMember(const char* key) const { return isMember(std::string_view(key)); }
    Json& member(std::string_view key);              // finds or inserts (one string is built, and only on insertion)
    Json& member(std::string&& key);                 // ... taking the key's storage on insertion
    Json const& member(std::string_view key) const;  // null value when absent
    Json& operator[](std::string const& key) { return member(std::string_view(key)); }
    Json& operator[](std::string&& key) { return member(std::move(key)); }
    Json& operator[](const char* key) { return member(std::string_view(key)); }
    Json const& operator[](std::string const& key) const { return member(std::string_view(key)); }
    Json const& operator[](const char* key) const { return member(std::string_view(key)); }
    void removeMember(std::string_view key)
    {
        if (kind_ == OBJECT)
        {
            auto it = v_.members->find(key);
            if (it != v_.members->end())
                v_.members->erase(it);
        }
    }
    Members const& members() const { return kind_ == OBJECT ? *v_.members : noMembers(); }
    std::vector<std::string> getMemberNames() const;

    // arrays
    size_t size() const { return kind_ == ARRAY ? v_.elements->size() : kind_ == OBJECT ? v_.members->size() : 0; }
    Json& append(Json v);
    Json& operator[](size_t i) { return elements().at(i); }
    Json const& operator[](size_t i) const { return elements().at(i); }
    Json& operator[](int i) { return elements().at((size_t)i); }
    Json const& operator[](int i) const { return elements().at((size_t)i); }
    Elements const& elements() const { return kind_ == ARRAY ? *v_.elements : noElements(); }
    Elements& elements();  // of an array; a null value becomes an empty array; anything else throws

    bool operator==(Json const& o) const;
    bool operator!=(Json const& o) const { return !(*this == o); }

private:
    void write(std::string& out, int indent, int depth) const;
    void destroy() noexcept;
    static Members const& noMembers();
    static Elements const& noElements();
    Kind kind_;
    union
    {
        int64_t i;
        uint64_t u;
        double d;
        std::string* s;
        Elements* elements;
        Members* members;
    } v_;
};
}  // namespace common
