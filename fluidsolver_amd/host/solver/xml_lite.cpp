#include "xml_lite.h"

#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

namespace xml_lite {

const Element *Element::first_child(const char *child_name) const {
    for (const auto &c : children)
        if (!child_name || c->name == child_name) return c.get();
    return nullptr;
}

const Element *Element::next_sibling(const char *sibling_name) const {
    if (!parent) return nullptr;
    bool past = false;
    for (const auto &c : parent->children) {
        if (past && (!sibling_name || c->name == sibling_name)) return c.get();
        if (c.get() == this) past = true;
    }
    return nullptr;
}

bool Element::int_attribute(const char *attr, int *out) const {
    for (const auto &a : attributes)
        if (a.first == attr) {
            char *end = nullptr;
            const long v = std::strtol(a.second.c_str(), &end, 10);
            if (end == a.second.c_str()) return false;
            *out = (int)v;
            return true;
        }
    return false;
}

namespace {

struct Parser {
    const std::string &s;
    size_t i = 0;
    std::string error;

    explicit Parser(const std::string &src) : s(src) {}

    bool starts(const char *lit) const { return s.compare(i, std::strlen(lit), lit) == 0; }
    void skip_space() {
        while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
    }
    bool skip_until(const char *lit) {
        const size_t p = s.find(lit, i);
        if (p == std::string::npos) return false;
        i = p + std::strlen(lit);
        return true;
    }
    static bool name_char(char c) {
        return std::isalnum((unsigned char)c) || c == '_' || c == '-' || c == ':' || c == '.';
    }
    std::string read_name() {
        const size_t b = i;
        while (i < s.size() && name_char(s[i])) i++;
        return s.substr(b, i - b);
    }
    static std::string unescape(const std::string &t) {
        std::string o;
        for (size_t k = 0; k < t.size(); k++) {
            if (t[k] == '&') {
                static const struct { const char *ent; char ch; } ents[] = {
                    {"&lt;", '<'}, {"&gt;", '>'}, {"&amp;", '&'}, {"&quot;", '"'}, {"&apos;", '\''}};
                bool hit = false;
                for (const auto &e : ents)
                    if (t.compare(k, std::strlen(e.ent), e.ent) == 0) {
                        o += e.ch;
                        k += std::strlen(e.ent) - 1;
                        hit = true;
                        break;
                    }
                if (hit) continue;
            }
            o += t[k];
        }
        return o;
    }

    // content of `parent` up to its closing tag (or to the end of input for the holder element)
    bool parse_content(Element *parent, bool top) {
        while (i < s.size()) {
            if (s[i] != '<') {
                const size_t b = i;
                while (i < s.size() && s[i] != '<') i++;
                if (!top) parent->text += unescape(s.substr(b, i - b));
                continue;
            }
            if (starts("<!--")) {
                if (!skip_until("-->")) return fail("unterminated comment");
                continue;
            }
            if (starts("<?")) {
                if (!skip_until("?>")) return fail("unterminated processing instruction");
                continue;
            }
            if (starts("<![CDATA[")) {
                const size_t b = i + 9;
                if (!skip_until("]]>")) return fail("unterminated CDATA");
                if (!top) parent->text += s.substr(b, i - 3 - b);
                continue;
            }
            if (starts("<!")) {
                if (!skip_until(">")) return fail("unterminated declaration");
                continue;
            }
            if (starts("</")) {
                if (top) return fail("closing tag without an open element");
                i += 2;
                const std::string n = read_name();
                skip_space();
                if (i >= s.size() || s[i] != '>') return fail("malformed closing tag");
                i++;
                if (n != parent->name) return fail(("closing tag </" + n + "> does not match <" + parent->name + ">").c_str());
                return true;
            }
            // an opening tag
            i++;
            auto el = std::make_unique<Element>();
            el->name = read_name();
            if (el->name.empty()) return fail("element name expected");
            el->parent = parent;
            bool self_closed = false;
            for (;;) {
                skip_space();
                if (i >= s.size()) return fail("unterminated tag");
                if (s[i] == '>') { i++; break; }
                if (starts("/>")) { i += 2; self_closed = true; break; }
                const std::string an = read_name();
                if (an.empty()) return fail("attribute name expected");
                skip_space();
                if (i >= s.size() || s[i] != '=') return fail("'=' expected after attribute name");
                i++;
                skip_space();
                if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) return fail("quoted attribute value expected");
                const char q = s[i++];
                const size_t b = i;
                while (i < s.size() && s[i] != q) i++;
                if (i >= s.size()) return fail("unterminated attribute value");
                el->attributes.emplace_back(an, unescape(s.substr(b, i - b)));
                i++;
            }
            Element *raw = el.get();
            parent->children.push_back(std::move(el));
            if (!self_closed && !parse_content(raw, false)) return false;
            size_t b = 0, e = raw->text.size();
            while (b < e && std::isspace((unsigned char)raw->text[b])) b++;
            while (e > b && std::isspace((unsigned char)raw->text[e - 1])) e--;
            raw->trimmed = raw->text.substr(b, e - b);
        }
        if (!top) return fail(("<" + parent->name + "> is never closed").c_str());
        return true;
    }
    bool fail(const char *what) {
        std::ostringstream o;
        o << what << " (offset " << i << ")";
        error = o.str();
        return false;
    }
};

// text of a leaf with leading white space removed (what tinyxml2 hands to sscanf / Query*Text)
const char *leaf_text(const Element *e) {
    if (!e) return nullptr;
    const char *t = e->text.c_str();
    while (*t && std::isspace((unsigned char)*t)) t++;
    return t;
}

}  // namespace

bool Document::parse(const std::string &source) {
    root_ = Element();
    Parser p(source);
    const bool ok = p.parse_content(&root_, true);
    error_ = ok ? "" : p.error;
    return ok;
}

bool Document::load_file(const char *path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) {
        error_ = std::string("cannot open ") + path;
        return false;
    }
    std::ostringstream buf;
    buf << f.rdbuf();
    return parse(buf.str());
}

void TagReader::Use(const Element *base) {
    base_ = base;
    if (!base) std::printf("WARNING: NULL Node Used.\n");
}

const char *TagReader::GetText(const char *name) const {
    if (!base_) return nullptr;
    const Element *e = base_->first_child(name);
    if (!e) return nullptr;
    return e->trimmed.empty() ? nullptr : e->trimmed.c_str();  // an empty tag has no text node
}

float TagReader::GetFloat(const char *name) const {
    float v = 0.f;
    if (base_)
        if (const char *t = leaf_text(base_->first_child(name))) std::sscanf(t, "%f", &v);
    return v;
}

int TagReader::GetInt(const char *name) const {
    int v = 0;
    if (base_)
        if (const char *t = leaf_text(base_->first_child(name))) std::sscanf(t, "%d", &v);
    return v;
}

cfloat3 TagReader::GetFloat3(const char *name) const {
    cfloat3 v(0.f, 0.f, 0.f);
    if (base_)
        if (const char *t = leaf_text(base_->first_child(name))) std::sscanf(t, "%f,%f,%f", &v.x, &v.y, &v.z);
    return v;
}

cint3 TagReader::GetInt3(const char *name) const {
    cint3 v(0, 0, 0);
    if (base_)
        if (const char *t = leaf_text(base_->first_child(name))) std::sscanf(t, "%d,%d,%d", &v.x, &v.y, &v.z);
    return v;
}

void TagReader::GetFloatN(float *buffer, int size, const char *name) const {
    if (!base_) return;
    const Element *e = base_->first_child(name);
    if (!e) {
        std::printf("Error: accessing invalid float n.\n");
        return;
    }
    const char *t = leaf_text(e);
    for (int k = 0; k < size; k++) {
        char *end = nullptr;
        const float v = std::strtof(t, &end);
        if (end == t) return;
        buffer[k] = v;
        t = end;
        while (*t && std::isspace((unsigned char)*t)) t++;
        if (*t != ',') return;
        t++;
    }
}

}  // namespace xml_lite
