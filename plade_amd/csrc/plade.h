// plade_amd/csrc/plade.h -- the C++ host API of PLADE, kept source compatible with the reference's
// code/PLADE/plade.h:44-96 (same four registration() overloads, same argument order and meaning,
// same bool/identity-on-failure behaviour, messages on std::cout/std::cerr), implemented on top of
// the C ABI of libplade_hip.so (include/plade_hip.h).
//
// The reference's signatures mention Eigen::Matrix<float,4,4>, pcl::PointCloud<pcl::PointNormal>::Ptr
// and PLANE (code/PLADE/plane_extraction.h:44-50).  This image has neither Eigen nor PCL/Boost, so
// "plade_compat.h" provides minimal stand-ins with the same names, layouts and members the
// signatures and the CLI use; define PLADE_USE_REAL_EIGEN_PCL before including this header in a
// tree that has the real libraries (INTEGRATION.md shows the reference-side change).
#ifndef PLADE_H
#define PLADE_H

#include <iosfwd>
#include <string>
#include <vector>

#ifdef PLADE_USE_REAL_EIGEN_PCL
#include <Eigen/Core>
#include <Eigen/LU>
#include <pcl/point_cloud.h>
#include <pcl/point_types.h>
#include "plane_extraction.h"
#else
#include "plade_compat.h"
#endif

/** plade.h:44-47 -- file names of two PLY point clouds. */
bool registration(Eigen::Matrix<float, 4, 4> &transformation,
                  const std::string &target_cloud_file,
                  const std::string &source_cloud_file);

/** plade.h:58-61 -- auto-tunes the RANSAC min support so that 10..40 planes are used. */
bool registration(Eigen::Matrix<float, 4, 4> &transformation,
                  pcl::PointCloud<pcl::PointNormal>::Ptr target_cloud,
                  pcl::PointCloud<pcl::PointNormal>::Ptr source_cloud);

/** plade.h:74-79 -- the user provides the extracted planes. */
bool registration(Eigen::Matrix<float, 4, 4> &transformation,
                  pcl::PointCloud<pcl::PointNormal>::Ptr target_cloud,
                  pcl::PointCloud<pcl::PointNormal>::Ptr source_cloud,
                  const std::vector<PLANE> &target_planes,
                  const std::vector<PLANE> &source_planes);

/** plade.h:91-96 -- explicit RANSAC min support per cloud. */
bool registration(Eigen::Matrix<float, 4, 4> &transformation,
                  pcl::PointCloud<pcl::PointNormal>::Ptr target_cloud,
                  pcl::PointCloud<pcl::PointNormal>::Ptr source_cloud,
                  int ransac_min_support_target,
                  int ransac_min_support_source);

/** Batch extension (no counterpart in the reference, whose batch mode is a plain loop of the file overload above,
 *  code/PLADE/main.cpp:122-148): `count` (1..registration_group_max = PLADE_GROUP_MAX) consecutive pairs of the list as ONE group.  Every pair gets the result,
 *  the messages and the identity-on-failure of the file overload -- its transformation is bit for bit the one the file overload
 *  returns -- but the plane extraction of all clouds of the group is one GPU launch sequence (plade_registration_pairs,
 *  include/plade_hip.h).  out[i] / err[i]: where pair i's console messages go (nullptr: std::cout / std::cerr). */
constexpr size_t registration_group_max = 8;
void registration_group(size_t count, Eigen::Matrix<float, 4, 4> *transformations, const std::string *target_cloud_files,
                        const std::string *source_cloud_files, bool *ok, std::ostream *const *out, std::ostream *const *err);

/** load_ply_cloud (code/PLADE/util.cpp:1505-1546): ascii / binary PLY with x y z nx ny nz. */
bool load_ply_cloud(const std::string &file_name, pcl::PointCloud<pcl::PointNormal> &cloud);

/** Select the GPU used by the calling thread's registrations (default 0). */
void plade_select_device(int device);
/** PLADE_TRACE_CLI=1: a time-stamped line on std::cerr (seconds since the first such call). */
void plade_cli_trace(const char *what);
/** GPUs this process sees (0 without one): the CLI's batch mode spreads the list over all of them by default. */
int plade_gpu_count();

/** Console of the calling thread's registrations: the messages the reference prints on std::cout / std::cerr go to
 *  these streams instead (nullptr = std::cout / std::cerr again).  Used by the CLI's batch workers. */
void plade_set_thread_console(std::ostream *out, std::ostream *err);

/** Destroy the calling thread's GPU context (it is created on first use and otherwise lives as long as the thread). */
void plade_release_thread_context();

#endif  // PLADE_H
