// host/obj_reader.hpp — OBJ scene load, same entry point as the reference
// (reference pathtracer/obj_reader.hpp:11): load_obj(path, out_vertices)
// appends three vertices per face.
#pragma once
#include <vector>

#include "bvh.hpp"

namespace hpt {
// Returns false when the file cannot be opened or holds a face this reader
// cannot parse.  Accepts what the reference accepts (`v x y z`, `f a b c`) and,
// beyond it, `a/b/c` index forms, negative indices and polygons (fan).
bool load_obj(const char *path, std::vector<Vec3f> &out_vertices);
}  // namespace hpt
