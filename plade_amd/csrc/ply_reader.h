// plade_amd/csrc/ply_reader.h -- PLY ingest for the CLI (C ABI: plade_ply_read).
// Reads the `vertex` element's floating-typed x y z (or X Y Z) and nx ny nz properties (ascii or binary LE/BE) into an
// interleaved float array, accepting, refusing and rounding exactly as the reference's ingest does (load_ply_cloud,
// code/PLADE/util.cpp:1505-1546, over code/PLADE/ply_reader.cpp and code/3rd_party/rply: see ply_reader.cpp); a binary file in
// the host's byte order whose vertex element is exactly `float x y z nx ny nz` (what PLADE's own sample data uses) is read
// with one bulk read instead of rply's per-value callbacks.
#pragma once
#include <functional>
#include <string>
#include <vector>

namespace plade {

// pos_nrm: N x 6.  Returns false (and fills err) on malformed files or when the vertex element has
// no complete position+normal set ("the number of points does not equal to the number of normals in
// the file", code/PLADE/util.cpp:1533-1536).
// before_grow: called just before `pos_nrm` has to be re-allocated (a caller that keeps the array page-locked between calls
// releases the registration of the old block there).
bool read_ply_pos_nrm(const std::string &path, std::vector<float> &pos_nrm, std::string &err,
                      std::vector<std::string> *warnings = nullptr, const std::function<void()> *before_grow = nullptr);

bool write_ply_pos_nrm(const std::string &path, const float *pos_nrm, size_t n);

}  // namespace plade
