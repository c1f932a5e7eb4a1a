// rt_leafword.h -- layout of a leaf's child word in the wide BVH, shared by its builder (rt_accel.h)
// and the kernels that search it (rt_kernels.h).
#pragma once
// A leaf's child word carries, above the primitive's index, what kind of path step its material
// asks for, so that a kernel sorting paths by that step (k_paths_pool) knows it the moment the
// closest hit is known, without reading the primitive and material records: bits 27-28 of the
// reference as stored in a wide node (private to the device; the descriptor's references have none).
#define RT4_CLASS_SHIFT 27u
#define RT4_CLASS_MASK 0x18000000u
#define RT4_CLASS_DIFFUSE 0u /* Lambertian, Isotropic: scatter through a pdf */
#define RT4_CLASS_METAL 1u
#define RT4_CLASS_GLASS 2u
#define RT4_CLASS_END 3u     /* DiffuseLight: the path ends there */


// A kind of leaf only the wide BVH has (the descriptor's references stop at RT_KIND_MEDIUM): six
// parallelograms of one list that are the faces of a box along the world's axes -- what box3 builds
// (hittable.rs:398-455) -- searched as one leaf with one slab test.  The index is into the box
// records the upload derives (rt_accel.h, WideBox); a hit is reported on the face's own quad.
#define RT_KIND_BOX 4u
