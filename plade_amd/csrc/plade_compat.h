// plade_amd/csrc/plade_compat.h -- minimal stand-ins for the Eigen / PCL types that appear in the
// reference's public API (code/PLADE/plade.h, plane_extraction.h:44-50) for builds without Eigen,
// PCL and Boost.  Only what the API and the CLI touch: a fixed 4x4 float matrix with Identity(),
// setIdentity(), operator()(r,c), inverse() and Eigen's default stream format; pcl::PointNormal with
// PCL's 48-byte layout; pcl::PointCloud<T> with points/size()/at()/push_back and a shared Ptr.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>
#include <iomanip>
#include <memory>
#include <ostream>
#include <sstream>
#include <string>
#include <vector>

// PLADE_USE_REAL_EIGEN: the real Eigen (header only; the reference vendors 3.4.0 under code/3rd_party/eigen-3.4.0) with
// the PCL stand-ins below -- what a tree without Boost can build, and what tests/test_host_logic.py compiles the host
// sources against to prove they are source compatible with the real Eigen::Matrix<float,4,4> / Vector3f.
#ifdef PLADE_USE_REAL_EIGEN
#include <Eigen/Core>
#include <Eigen/LU>
#else
namespace Eigen {

template <typename Scalar, int Rows, int Cols>
class Matrix;

template <>
class Matrix<float, 4, 4> {
public:
    Matrix() { std::memset(m_, 0, sizeof(m_)); }
    static Matrix Identity() { Matrix r; r.setIdentity(); return r; }
    void setIdentity() {
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) m_[c][r] = r == c ? 1.f : 0.f;
    }
    float &operator()(int r, int c) { return m_[c][r]; }          // column-major like Eigen's default
    const float &operator()(int r, int c) const { return m_[c][r]; }
    const float *data() const { return &m_[0][0]; }
    float *data() { return &m_[0][0]; }
    // general 4x4 inverse by cofactors in double (plade.cpp:704 calls Matrix4f::inverse())
    Matrix inverse() const {
        double a[16], inv[16];
        for (int r = 0; r < 4; ++r) for (int c = 0; c < 4; ++c) a[4 * r + c] = (*this)(r, c);
        inv[0] = a[5] * a[10] * a[15] - a[5] * a[11] * a[14] - a[9] * a[6] * a[15] + a[9] * a[7] * a[14] + a[13] * a[6] * a[11] - a[13] * a[7] * a[10];
        inv[4] = -a[4] * a[10] * a[15] + a[4] * a[11] * a[14] + a[8] * a[6] * a[15] - a[8] * a[7] * a[14] - a[12] * a[6] * a[11] + a[12] * a[7] * a[10];
        inv[8] = a[4] * a[9] * a[15] - a[4] * a[11] * a[13] - a[8] * a[5] * a[15] + a[8] * a[7] * a[13] + a[12] * a[5] * a[11] - a[12] * a[7] * a[9];
        inv[12] = -a[4] * a[9] * a[14] + a[4] * a[10] * a[13] + a[8] * a[5] * a[14] - a[8] * a[6] * a[13] - a[12] * a[5] * a[10] + a[12] * a[6] * a[9];
        inv[1] = -a[1] * a[10] * a[15] + a[1] * a[11] * a[14] + a[9] * a[2] * a[15] - a[9] * a[3] * a[14] - a[13] * a[2] * a[11] + a[13] * a[3] * a[10];
        inv[5] = a[0] * a[10] * a[15] - a[0] * a[11] * a[14] - a[8] * a[2] * a[15] + a[8] * a[3] * a[14] + a[12] * a[2] * a[11] - a[12] * a[3] * a[10];
        inv[9] = -a[0] * a[9] * a[15] + a[0] * a[11] * a[13] + a[8] * a[1] * a[15] - a[8] * a[3] * a[13] - a[12] * a[1] * a[11] + a[12] * a[3] * a[9];
        inv[13] = a[0] * a[9] * a[14] - a[0] * a[10] * a[13] - a[8] * a[1] * a[14] + a[8] * a[2] * a[13] + a[12] * a[1] * a[10] - a[12] * a[2] * a[9];
        inv[2] = a[1] * a[6] * a[15] - a[1] * a[7] * a[14] - a[5] * a[2] * a[15] + a[5] * a[3] * a[14] + a[13] * a[2] * a[7] - a[13] * a[3] * a[6];
        inv[6] = -a[0] * a[6] * a[15] + a[0] * a[7] * a[14] + a[4] * a[2] * a[15] - a[4] * a[3] * a[14] - a[12] * a[2] * a[7] + a[12] * a[3] * a[6];
        inv[10] = a[0] * a[5] * a[15] - a[0] * a[7] * a[13] - a[4] * a[1] * a[15] + a[4] * a[3] * a[13] + a[12] * a[1] * a[7] - a[12] * a[3] * a[5];
        inv[14] = -a[0] * a[5] * a[14] + a[0] * a[6] * a[13] + a[4] * a[1] * a[14] - a[4] * a[2] * a[13] - a[12] * a[1] * a[6] + a[12] * a[2] * a[5];
        inv[3] = -a[1] * a[6] * a[11] + a[1] * a[7] * a[10] + a[5] * a[2] * a[11] - a[5] * a[3] * a[10] - a[9] * a[2] * a[7] + a[9] * a[3] * a[6];
        inv[7] = a[0] * a[6] * a[11] - a[0] * a[7] * a[10] - a[4] * a[2] * a[11] + a[4] * a[3] * a[10] + a[8] * a[2] * a[7] - a[8] * a[3] * a[6];
        inv[11] = -a[0] * a[5] * a[11] + a[0] * a[7] * a[9] + a[4] * a[1] * a[11] - a[4] * a[3] * a[9] - a[8] * a[1] * a[7] + a[8] * a[3] * a[5];
        inv[15] = a[0] * a[5] * a[10] - a[0] * a[6] * a[9] - a[4] * a[1] * a[10] + a[4] * a[2] * a[9] + a[8] * a[1] * a[6] - a[8] * a[2] * a[5];
        double det = a[0] * inv[0] + a[1] * inv[4] + a[2] * inv[8] + a[3] * inv[12];
        Matrix r;
        det = 1.0 / det;
        for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) r(i, j) = (float)(inv[4 * i + j] * det);
        return r;
    }

private:
    float m_[4][4];
};

// Eigen's default IOFormat (Eigen/src/Core/IO.h): StreamPrecision, columns aligned to the widest
// entry of the whole matrix, " " between coefficients, "\n" between rows; this is what
// `output << transformation` in code/PLADE/main.cpp:84-86 writes.
inline std::ostream &operator<<(std::ostream &s, const Matrix<float, 4, 4> &m) {
    std::streamsize width = 0;
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) {
            std::stringstream sstr;
            sstr.copyfmt(s);
            sstr << m(r, c);
            width = std::max<std::streamsize>(width, (std::streamsize)sstr.str().length());
        }
    const std::streamsize old_width = s.width();
    const char old_fill = s.fill();
    for (int r = 0; r < 4; ++r) {
        if (r) s << "\n";
        if (width) { s.fill(' '); s.width(width); }
        s << m(r, 0);
        for (int c = 1; c < 4; ++c) {
            s << " ";
            if (width) { s.fill(' '); s.width(width); }
            s << m(r, c);
        }
    }
    if (width) { s.fill(old_fill); s.width(old_width); }
    return s;
}

template <typename Scalar, int Rows, int Cols>
class Matrix {  // only Vector3f is needed (PLANE::normal)
public:
    Matrix() { for (int i = 0; i < Rows * Cols; ++i) v_[i] = Scalar(0); }
    Matrix(Scalar a, Scalar b, Scalar c) { static_assert(Rows * Cols == 3, "3-vector ctor"); v_[0] = a; v_[1] = b; v_[2] = c; }
    Scalar &operator[](int i) { return v_[i]; }
    const Scalar &operator[](int i) const { return v_[i]; }
    Scalar x() const { return v_[0]; }
    Scalar y() const { return v_[1]; }
    Scalar z() const { return v_[2]; }

private:
    Scalar v_[Rows * Cols];
};
typedef Matrix<float, 3, 1> Vector3f;
typedef Matrix<float, 4, 4> Matrix4f;

}  // namespace Eigen
#endif  // PLADE_USE_REAL_EIGEN

namespace pcl {

// pcl::PointNormal: x y z (pad) | normal_x normal_y normal_z (pad) | curvature (pad x3) = 48 bytes
struct alignas(16) PointNormal {
    float x, y, z, data_pad;
    float normal_x, normal_y, normal_z, normal_pad;
    float curvature, pad2[3];
    PointNormal() : x(0), y(0), z(0), data_pad(1.f), normal_x(0), normal_y(0), normal_z(0), normal_pad(0), curvature(0) {
        pad2[0] = pad2[1] = pad2[2] = 0;
    }
    PointNormal(float px, float py, float pz, float nx, float ny, float nz)
        : x(px), y(py), z(pz), data_pad(1.f), normal_x(nx), normal_y(ny), normal_z(nz), normal_pad(0), curvature(0) {
        pad2[0] = pad2[1] = pad2[2] = 0;
    }
};
static_assert(sizeof(PointNormal) == 48, "pcl::PointNormal layout");

template <typename PointT>
class PointCloud {
public:
    typedef std::shared_ptr<PointCloud<PointT>> Ptr;            // boost::shared_ptr in PCL 1.8.1
    typedef std::shared_ptr<const PointCloud<PointT>> ConstPtr;
    std::vector<PointT> points;
    uint32_t width = 0, height = 1;
    bool is_dense = true;
    size_t size() const { return points.size(); }
    bool empty() const { return points.empty(); }
    void resize(size_t n) { points.resize(n); width = (uint32_t)n; }
    void reserve(size_t n) { points.reserve(n); }
    void clear() { points.clear(); width = 0; }
    void push_back(const PointT &p) { points.push_back(p); width = (uint32_t)points.size(); }
    PointT &at(size_t i) { return points.at(i); }
    const PointT &at(size_t i) const { return points.at(i); }
    PointT &operator[](size_t i) { return points[i]; }
    const PointT &operator[](size_t i) const { return points[i]; }
};

}  // namespace pcl

// code/PLADE/plane_extraction.h:44-50
class PLANE : public std::vector<int> {
public:
    PLANE() {}
    template <class InputIt>
    PLANE(InputIt first, InputIt last) : std::vector<int>(first, last) {}
    Eigen::Vector3f normal;
    float d = 0.f;
};

// code/PLADE/plane_extraction.h:56-63
class PlaneExtraction {
public:
    static std::vector<PLANE> detect(const pcl::PointCloud<pcl::PointNormal> &cloud, unsigned int min_support = 1000,
                                     float dist_thresh = 0.005f, float bitmap_reso = 0.02f, float normal_thresh = 0.8f,
                                     float overlook_prob = 0.001f);
};
