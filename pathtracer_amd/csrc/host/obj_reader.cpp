// host/obj_reader.cpp — line-based OBJ reader (see obj_reader.hpp).
// The reference tokenises with fscanf("%s") and prints every vertex
// (reference pathtracer/obj_reader.cpp:20-75); this one reads whole lines,
// prints nothing, and yields the same vertex list on files the reference can read.
#include "obj_reader.hpp"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace hpt {

bool load_obj(const char *path, std::vector<Vec3f> &out_vertices) {
    FILE *f = std::fopen(path, "r");
    if (!f) return false;
    std::vector<Vec3f> verts;
    std::vector<long> face;
    char line[4096];
    bool ok = true;
    while (ok && std::fgets(line, sizeof line, f)) {
        char *p = line;
        while (*p == ' ' || *p == '\t') p++;
        if (p[0] == 'v' && (p[1] == ' ' || p[1] == '\t')) {
            Vec3f v;
            char *q = p + 1;
            for (int k = 0; k < 3; k++) v[k] = std::strtof(q, &q);
            verts.push_back(v);
        } else if (p[0] == 'f' && (p[1] == ' ' || p[1] == '\t')) {
            face.clear();
            char *q = p + 1;
            for (;;) {
                while (*q == ' ' || *q == '\t') q++;
                if (*q == '\0' || *q == '\n' || *q == '\r' || *q == '#') break;
                char *e;
                long idx = std::strtol(q, &e, 10);
                if (e == q) { ok = false; break; }
                while (*e && *e != ' ' && *e != '\t' && *e != '\n' && *e != '\r') e++;   // skip /vt/vn
                q = e;
                if (idx < 0) idx = (long)verts.size() + 1 + idx;
                if (idx < 1 || idx > (long)verts.size()) { ok = false; break; }
                face.push_back(idx - 1);
            }
            if (ok && face.size() < 3) ok = false;
            for (size_t k = 2; ok && k < face.size(); k++) {
                out_vertices.push_back(verts[face[0]]);
                out_vertices.push_back(verts[face[k - 1]]);
                out_vertices.push_back(verts[face[k]]);
            }
        }
    }
    std::fclose(f);
    return ok;
}

}  // namespace hpt
