// A small read-only XML tree, enough for the solver's config files (several top-level elements,
// attributes, comments, text leaves).  The reference reads the same files through tinyxml2 and its
// Tinyxml_Reader wrapper (catpaw/cpXMLHelper.cpp:5-85 under /root/reference/particle_solver); the
// TagReader below keeps that wrapper's conventions: a missing tag reads as 0, vectors are
// "a, b, c" parsed with sscanf("%f,%f,%f").
#pragma once

#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../catpaw/geometry.h"

namespace xml_lite {

struct Element {
    std::string name;
    std::string text;     // concatenated character data directly inside this element
    std::string trimmed;  // the same without leading and trailing white space
    std::vector<std::pair<std::string, std::string>> attributes;
    std::vector<std::unique_ptr<Element>> children;
    Element *parent = nullptr;

    const Element *first_child(const char *child_name = nullptr) const;
    // next element after this one under the same parent, optionally of a given name
    const Element *next_sibling(const char *sibling_name = nullptr) const;
    bool int_attribute(const char *attr, int *out) const;
};

class Document {
public:
    // false (and an explanation in error()) when the file cannot be read or is malformed
    bool load_file(const char *path);
    bool parse(const std::string &source);
    const Element *first_child(const char *name = nullptr) const { return root_.first_child(name); }
    const std::string &error() const { return error_; }

private:
    Element root_;  // unnamed holder of the top-level elements
    std::string error_;
};

// Reads typed values from the children of one element.
class TagReader {
public:
    void Use(const Element *base);
    const char *GetText(const char *name) const;  // nullptr when the tag is missing
    float GetFloat(const char *name) const;
    int GetInt(const char *name) const;
    cfloat3 GetFloat3(const char *name) const;
    cint3 GetInt3(const char *name) const;
    // comma separated list; reads at most `size` values, leaves the rest of `buffer` untouched
    void GetFloatN(float *buffer, int size, const char *name) const;

private:
    const Element *base_ = nullptr;
};

}  // namespace xml_lite
