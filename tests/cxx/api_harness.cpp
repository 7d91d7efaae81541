// tests/cxx/api_harness.cpp -- drives the C++ host API of plade_amd/csrc/plade.h the way a user of the
// reference's plade.h would: the four registration() overloads + PlaneExtraction::detect.
// Usage: api_harness target.ply source.ply   (prints one block per overload)
#include "plade.h"

#include <iostream>

static void show(const char *tag, bool ok, const Eigen::Matrix<float, 4, 4> &T) {
    std::cout << "@" << tag << " " << (ok ? 1 : 0) << "\n";
    std::cout.precision(9);
    for (int r = 0; r < 4; ++r) {   // the library reports progress on stdout like the reference: rows are tagged
        std::cout << "@row";
        for (int c = 0; c < 4; ++c) std::cout << " " << T(r, c);
        std::cout << std::endl;
    }
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    Eigen::Matrix<float, 4, 4> T;
    show("files", registration(T, argv[1], argv[2]), T);
    pcl::PointCloud<pcl::PointNormal>::Ptr tgt(new pcl::PointCloud<pcl::PointNormal>), src(new pcl::PointCloud<pcl::PointNormal>);
    if (!load_ply_cloud(argv[1], *tgt) || !load_ply_cloud(argv[2], *src)) return 3;
    std::cout << "@sizes " << tgt->size() << " " << src->size() << "\n";
    show("clouds", registration(T, tgt, src), T);
    show("minsupport", registration(T, tgt, src, 1500, 1500), T);
    std::vector<PLANE> tp = PlaneExtraction::detect(*tgt, 1500), sp = PlaneExtraction::detect(*src, 1500);
    std::cout << "@planes " << tp.size() << " " << sp.size() << "\n";
    show("planes", registration(T, tgt, src, tp, sp), T);
    // failure path: no planes -> false, matrix untouched by the planes overload contract = identity here
    std::vector<PLANE> none;
    Eigen::Matrix<float, 4, 4> I = Eigen::Matrix<float, 4, 4>::Identity();
    show("noplanes", registration(I, tgt, src, none, none), I);
    return 0;
}
