// plade_amd/csrc/ply_reader.h -- PLY ingest for the CLI.
// Reads the `vertex` element's x y z nx ny nz properties (any scalar type, ascii or binary LE/BE)
// into an interleaved float array; a binary little-endian file whose vertex element is exactly
// `float x y z nx ny nz` (what PLADE's own sample data uses) is read with one bulk read instead of
// rply's per-value callbacks (code/PLADE/ply_reader.cpp:60-93, code/3rd_party/rply).
#pragma once
#include <string>
#include <vector>

namespace plade {

// pos_nrm: N x 6.  Returns false (and fills err) on malformed files or when the vertex element has
// no complete position+normal set ("the number of points does not equal to the number of normals in
// the file", code/PLADE/util.cpp:1533-1536).
bool read_ply_pos_nrm(const std::string &path, std::vector<float> &pos_nrm, std::string &err,
                      std::vector<std::string> *warnings = nullptr);

bool write_ply_pos_nrm(const std::string &path, const float *pos_nrm, size_t n);

}  // namespace plade
