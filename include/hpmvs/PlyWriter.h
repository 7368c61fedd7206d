// Extended PLY output of refined patches: the vertex + point_visibility layout of the reference's
// DynOctTree::toExtPly (reference include/hpmvs/doctree.h:526-622): x y z [nx ny nz] red green blue
// [scalar_scale] per vertex, then one "list uint uint visible_cameras" per patch.  The reference
// walks its octree leaves; here the caller hands over the patch list.
#ifndef HPMVS_PLYWRITER_H_
#define HPMVS_PLYWRITER_H_
#include <vector>
#include <hpmvs/Patch3d.h>
namespace mo3d {
bool writeExtPly(const char* name, const std::vector<Ppatch3d>& patches, bool binary = true, bool normal = true,
                 bool scale = true, bool visibility = true);
}
#endif
