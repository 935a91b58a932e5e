// Test-only declaration of the reference's device-backend interface: the constructor and the nine public methods of
// class OpenCL_Manager (reference pathtracer/opencl_manager.hpp:11-50) with the argument types they name, so that
// the forwarding body printed in INTEGRATION.md can be compiled (tests/test_host.py::
// test_integration_forwarding_body_compiles).  Boundary check only: it pins that the replacement body matches the
// interface the reference's Pathtracer calls, nothing about parity.  The PODs are declared as far as the body
// needs them (it only passes their addresses on; Pixel's size is used).
#pragma once
#include <string>

struct cl_float4 { float s[4]; };
struct Pixel { float red, green, blue; };
struct Triangle; struct BVH_Node_BBox; struct BVH_Node_Info; struct BVH_Node; struct Primitive; struct Material; struct Camera;

class OpenCL_Manager {
public:
    OpenCL_Manager();                       // opencl_manager.hpp:11, defined opencl_manager.cpp:42, called pathtracer.cpp:69
    int initialize(int width, int height);
    int kernel_load(std::string file_name, std::string kernel_name);
    int new_kernel_load(int kid, std::string file_name, std::string kernel_name);
    int triangles_to_texture(Triangle *triangles, int n_triangles);
    int bvh_to_texture(BVH_Node_BBox *bbox, BVH_Node_Info *info, int n_nodes);
    int kernel_execute(Primitive *primitives, Camera *camera, int n_primitives, Pixel *image, int width, int height,
                       int *seed_memory, int iterations, Triangle *triangles, int n_triangles, Material *materials,
                       int n_materials, BVH_Node *nodes, int n_nodes);
    int ray_kernel_execute(Primitive *primitives, Camera *camera, int n_primitives, int width, int height, int *seed_memory,
                           int iterations, Triangle *triangles, int n_triangles, Material *materials, int n_materials,
                           int n_nodes, cl_float4 *color, cl_float4 *origin, cl_float4 *direction, cl_float4 *weight,
                           int *bounce, cl_float4 *first_color, cl_float4 *first_origin, cl_float4 *first_direction,
                           cl_float4 *first_weight, int *iteration_memory);
    int initial_ray_kernel_execute(Primitive *primitives, Camera *camera, int n_primitives, int width, int height,
                                   int *seed_memory, int iterations, Triangle *triangles, int n_triangles,
                                   Material *materials, int n_materials, BVH_Node *nodes, int n_nodes, cl_float4 *color,
                                   cl_float4 *origin, cl_float4 *direction, cl_float4 *weight);
    int reconstruct_kernel_execute(Pixel *image, int dim, cl_float4 *contribution, int *iteration_memory);
};
