// PLY vertex reader of the host mirror: what DepthNormInit::readDepths / readNormals get from io/io_file.c's
// ply_header_read + ply_read_1 over RPly (depth_normal_init.cpp:94-134, io/io_file.c:20-128): the x y z (and, when asked
// for and present, nx ny nz) of every vertex as doubles, DIM = 3 per vertex.
#pragma once
#include <string>
#include <vector>

namespace mvshost {

// 0 on success.  points: 3 doubles per vertex; normals (may be null): filled, 3 per vertex, only if the file has nx ny nz --
// left empty otherwise (ply_read_1 sets no callback for them then, io_file.c:84-90).
int readPlyVertices(const std::string& file, std::vector<double>& points, std::vector<double>* normals, std::string* error = nullptr);

}  // namespace mvshost
