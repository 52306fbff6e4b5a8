#include "scene_setup.h"

#include <cstdio>
#include <cstring>

using xml_lite::Element;
using xml_lite::TagReader;

void loadFluidVolume(const Element *scene, int typenum, std::vector<fluidvol> &out) {
    if (!scene) {
        std::printf("error loading fluid volume: invalid scene element.\n");
        return;
    }
    TagReader reader;
    for (const Element *e = scene->first_child("Vol"); e && e->name == "Vol"; e = e->next_sibling()) {
        reader.Use(e);
        fluidvol fv;
        fv.xmin = reader.GetFloat3("VolMin");
        fv.xmax = reader.GetFloat3("VolMax");
        fv.group = reader.GetInt("Group");
        const char *kind = reader.GetText("Type");
        const char *hollow = reader.GetText("Empty");
        if (!kind || std::strcmp(kind, "fluid") == 0) fv.type = TYPE_FLUID;
        else if (std::strcmp(kind, "deformable") == 0) fv.type = TYPE_DEFORMABLE;
        else if (std::strcmp(kind, "granular") == 0) fv.type = TYPE_GRANULAR;
        else if (std::strcmp(kind, "cloth") == 0) fv.type = TYPE_CLOTH;
        fv.empty = hollow && std::strcmp(hollow, "false") != 0;
        reader.GetFloatN(fv.volfrac, typenum, "VolFrac");
        out.push_back(fv);
    }
}

void fill_volume_points(const fluidvol &fv, float spacing, std::vector<cfloat3> &points) {
    const cfloat3 lo = fv.xmin, hi = fv.xmax;
    if (!fv.empty) {
        for (float x = lo.x; x < hi.x; x += spacing)
            for (float y = lo.y; y < hi.y; y += spacing)
                for (float z = lo.z; z < hi.z; z += spacing) points.push_back(cfloat3(x, y, z));
        return;
    }
    for (float x = lo.x; x < hi.x; x += spacing)
        for (float y = lo.y; y < hi.y; y += spacing) {
            points.push_back(cfloat3(x, y, lo.z));
            points.push_back(cfloat3(x, y, hi.z));
        }
    for (float x = lo.x; x < hi.x; x += spacing)
        for (float z = lo.z; z < hi.z; z += spacing) {
            points.push_back(cfloat3(x, lo.y, z));
            points.push_back(cfloat3(x, hi.y, z));
        }
    for (float y = lo.y; y < hi.y; y += spacing)
        for (float z = lo.z; z < hi.z; z += spacing) {
            points.push_back(cfloat3(lo.x, y, z));
            points.push_back(cfloat3(hi.x, y, z));
        }
}

namespace {

void emit(ParticleObject *o, const cfloat3 &p, const cfloat3 &n, int id) {
    o->pos.push_back(p);
    o->normal.push_back(n);
    o->type.push_back(TYPE_RIGID);
    o->id.push_back(id);
}

// the walls of a box as pairs of opposite faces; `lid` = with the y = ymax face
void box_faces(ParticleObject *o, const cfloat3 &lo, const cfloat3 &hi, float spacing, int id, bool lid) {
    for (float x = lo.x; x <= hi.x; x += spacing)
        for (float y = lo.y; y <= hi.y; y += spacing) {
            emit(o, cfloat3(x, y, lo.z), cfloat3(0, 0, 1), id);
            emit(o, cfloat3(x, y, hi.z), cfloat3(0, 0, -1), id);
        }
    for (float z = lo.z; z <= hi.z; z += spacing)
        for (float y = lo.y; y <= hi.y; y += spacing) {
            emit(o, cfloat3(lo.x, y, z), cfloat3(1, 0, 0), id);
            emit(o, cfloat3(hi.x, y, z), cfloat3(-1, 0, 0), id);
        }
    for (float x = lo.x; x <= hi.x; x += spacing)
        for (float z = lo.z; z <= hi.z; z += spacing) {
            emit(o, cfloat3(x, lo.y, z), cfloat3(0, 1, 0), id);
            if (lid) emit(o, cfloat3(x, hi.y, z), cfloat3(0, -1, 0), id);
        }
}

}  // namespace

void BoundaryGenerator::parseNode(const Element *e, ParticleObject *out) {
    TagReader reader;
    reader.Use(e);
    const cfloat3 lo = reader.GetFloat3("xmin"), hi = reader.GetFloat3("xmax");
    const float spacing = reader.GetFloat("spacing");
    const int id = reader.GetInt("id");
    if (!(spacing > 0.f)) {
        std::printf("boundary shape <%s> without a positive <spacing>: skipped\n", e->name.c_str());
        return;
    }
    if (e->name == "Cube") {
        const cfloat3 normal = reader.GetFloat3("normal");
        for (float x = lo.x; x <= hi.x; x += spacing)
            for (float y = lo.y; y <= hi.y; y += spacing)
                for (float z = lo.z; z <= hi.z; z += spacing) emit(out, cfloat3(x, y, z), normal, id);
    } else if (e->name == "Box") {
        box_faces(out, lo, hi, spacing, id, true);
    } else if (e->name == "OpenBox") {
        box_faces(out, lo, hi, spacing, id, false);
    }
}

ParticleObject *BoundaryGenerator::loadxml(const char *filepath) {
    std::printf("Parsing XML.\n");
    xml_lite::Document doc;
    if (!doc.load_file(filepath)) {
        std::printf("boundary file %s: %s\n", filepath, doc.error().c_str());
        return nullptr;
    }
    const Element *root = doc.first_child("Boundary");
    if (!root) {
        std::printf("boundary file %s has no <Boundary> element\n", filepath);
        return nullptr;
    }
    ParticleObject *so = new ParticleObject();
    for (const Element *c = root->first_child(); c; c = c->next_sibling()) parseNode(c, so);
    return so;
}
