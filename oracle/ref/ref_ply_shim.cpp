// oracle/ref/ref_ply_shim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// extern-"C" shim over the reference's PLY ingest, compiled where it lies: code/3rd_party/rply/rply.c (the parser) and
// code/PLADE/ply_reader.cpp (PlyReader::read :46-152, collect_elements :277-386) are built from /root/reference by
// oracle/ref/Makefile; the one function of the ingest that cannot be compiled here is its last caller, load_ply_cloud
// (code/PLADE/util.cpp:1505-1546), because util.cpp pulls in PCL (Boost) for the output container.  ref_ply_read below is
// OUR restatement of those 40 lines over a plain float array -- same element / property selection, same failure conditions,
// same console warnings -- citing them line by line.  Never linked into, imported by or called from the product.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "ply_reader.h"     // the reference's, -I$(REF)/code/PLADE

extern "C" {

// load_ply_cloud (util.cpp:1505-1546).  Returns 1 and a malloc'ed n x 6 array (x y z nx ny nz) on success, 0 when the
// reference's function returns false (*out = nullptr, *n = 0).  The caller frees with ref_ply_free.
int ref_ply_read(const char *path, float **out, long *n) {
    *out = nullptr; *n = 0;
    std::vector<Element> elements;
    PlyReader reader;
    if (!reader.read(path, elements)) return 0;                              // util.cpp:1508-1509
    std::vector<float> cloud;                                                // cloud.resize(...) of :1511-1517 only sizes the container
    size_t cloud_size = 0;
    for (std::size_t i = 0; i < elements.size(); ++i) {
        const Element &e = elements[i];
        if (e.name == "vertex") { cloud_size = e.num_instances; break; }     // util.cpp:1513-1516
    }
    for (std::size_t i = 0; i < elements.size(); ++i) {                      // util.cpp:1519-1543
        const Element &e = elements[i];
        if (e.name == "vertex") {
            std::vector<vec3> points, normals;
            for (const auto &p : e.vec3_properties) {
                std::string name = p.name;
                if (name.find("point") != std::string::npos) points = p;
                else if (name.find("normal") != std::string::npos) normals = p;
                else std::cout << "Warning: ignored property '" << name << "'" << std::endl;
            }
            if (points.size() != normals.size()) {                           // util.cpp:1533-1536
                std::cerr << "the number of points does not equal to the number of normals in the file" << std::endl;
                return 0;
            }
            cloud.resize(6 * points.size());                                 // util.cpp:1537-1539
            cloud_size = points.size();
            for (std::size_t j = 0; j < points.size(); ++j) {
                cloud[6 * j] = points[j].x; cloud[6 * j + 1] = points[j].y; cloud[6 * j + 2] = points[j].z;
                cloud[6 * j + 3] = normals[j].x; cloud[6 * j + 4] = normals[j].y; cloud[6 * j + 5] = normals[j].z;
            }
        } else
            std::cout << "Warning: unknown element '" << e.name << std::endl;
    }
    if (!(cloud_size > 0)) return 0;                                         // util.cpp:1545: return cloud.size() > 0
    if (cloud.size() != 6 * cloud_size) cloud.resize(6 * cloud_size, 0.f);   // (a vertex element found by :1513 but not by :1521 cannot happen)
    float *o = static_cast<float *>(malloc(cloud.size() * sizeof(float)));
    if (!o) return 0;
    memcpy(o, cloud.data(), cloud.size() * sizeof(float));
    *out = o; *n = (long)cloud_size;
    return 1;
}

void ref_ply_free(float *p) { free(p); }

}  // extern "C"
