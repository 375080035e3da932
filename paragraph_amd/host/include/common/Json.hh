// Small JSON document model for the host workflow: graph descriptions in (share/schema/graph_schema.json), count / genotype
// documents out (share/schema/output_schema.json).  Stands where the reference uses jsoncpp's Json::Value; objects keep
// their keys sorted like Json::Value does, so documents written from here list members in the same order.
#pragma once
#include <cstdint>
#include <map>
#include <string>
#include <vector>

namespace common
{
class Json
{
public:
    enum Kind { NUL, BOOL, INT, UINT, REAL, STRING, ARRAY, OBJECT };
    typedef std::map<std::string, Json> Members;
    typedef std::vector<Json> Elements;

    Json() = default;
    Json(bool v) : kind_(BOOL), u_(v) {}
    Json(int v) : kind_(INT), i_(v) {}
    Json(int64_t v) : kind_(INT), i_(v) {}
    Json(unsigned v) : kind_(UINT), u_(v) {}
    Json(uint64_t v) : kind_(UINT), u_(v) {}
    Json(double v) : kind_(REAL), d_(v) {}
    Json(const char* v) : kind_(STRING), s_(v) {}
    Json(std::string v) : kind_(STRING), s_(std::move(v)) {}
    static Json array() { Json j; j.kind_ = ARRAY; return j; }
    static Json object() { Json j; j.kind_ = OBJECT; return j; }
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
    bool isMember(std::string const& key) const { return kind_ == OBJECT && members_.count(key) != 0; }
    Json& operator[](std::string const& key);
    Json& operator[](const char* key) { return (*this)[std::string(key)]; }
    Json const& operator[](std::string const& key) const;  // null value when absent
    Json const& operator[](const char* key) const { return (*this)[std::string(key)]; }
    void removeMember(std::string const& key) { members_.erase(key); }
    Members const& members() const { return members_; }
    std::vector<std::string> getMemberNames() const;

    // arrays
    size_t size() const { return kind_ == ARRAY ? elements_.size() : kind_ == OBJECT ? members_.size() : 0; }
    Json& append(Json v);
    Json& operator[](size_t i) { return elements_.at(i); }
    Json const& operator[](size_t i) const { return elements_.at(i); }
    Json& operator[](int i) { return elements_.at((size_t)i); }
    Json const& operator[](int i) const { return elements_.at((size_t)i); }
    Elements const& elements() const { return elements_; }
    Elements& elements() { return elements_; }

    bool operator==(Json const& o) const;
    bool operator!=(Json const& o) const { return !(*this == o); }

private:
    void write(std::string& out, int indent, int depth) const;
    Kind kind_ = NUL;
    int64_t i_ = 0;
    uint64_t u_ = 0;
    double d_ = 0;
    std::string s_;
    Elements elements_;
    Members members_;
};
}  // namespace common
