// mini_eigen.hpp — the handful of Eigen types the FastDEM public API mentions, for hosts
// without Eigen3 (this image has none).  With Eigen installed this header is a no-op and the
// real types are used; layouts match (column-major, Isometry3d::matrix().data() = double[16]).
#pragma once
#if __has_include(<Eigen/Geometry>)
#include <Eigen/Core>
#include <Eigen/Geometry>
#else
#include <array>
#include <cmath>
#include <cstddef>

namespace Eigen {

template <typename T, int N>
struct Vec {
  std::array<T, N> v{};
  Vec() = default;
  Vec(T a, T b) { static_assert(N == 2, ""); v = {a, b}; }
  Vec(T a, T b, T c) { static_assert(N == 3, ""); v = {a, b, c}; }
  T& operator()(int i) { return v[size_t(i)]; }
  T operator()(int i) const { return v[size_t(i)]; }
  T& operator[](int i) { return v[size_t(i)]; }
  T operator[](int i) const { return v[size_t(i)]; }
  T& x() { return v[0]; }
  T x() const { return v[0]; }
  T& y() { return v[1]; }
  T y() const { return v[1]; }
  T& z() { static_assert(N >= 3, ""); return v[2]; }
  T z() const { static_assert(N >= 3, ""); return v[2]; }
  T* data() { return v.data(); }
  const T* data() const { return v.data(); }
  static Vec Zero() { return Vec(); }
  static Vec UnitX() { Vec r; r.v[0] = T(1); return r; }
  static Vec UnitY() { Vec r; r.v[1] = T(1); return r; }
  static Vec UnitZ() { static_assert(N >= 3, ""); Vec r; r.v[2] = T(1); return r; }
  template <int M = N>
  Vec<T, 2> head() const { static_assert(M >= 2, ""); return Vec<T, 2>(v[0], v[1]); }
  Vec operator+(const Vec& o) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] + o.v[i]; return r; }
  Vec operator-(const Vec& o) const { Vec r; for (int i = 0; i < N; ++i) r.v[i] = v[i] - o.v[i]; return r; }
  T squaredNorm() const { T s = 0; for (int i = 0; i < N; ++i) s += v[i] * v[i]; return s; }
  T norm() const { return std::sqrt(squaredNorm()); }
  template <typename U> Vec<U, N> cast() const { Vec<U, N> r; for (int i = 0; i < N; ++i) r.v[i] = U(v[i]); return r; }
};
using Vector2d = Vec<double, 2>;
using Vector3d = Vec<double, 3>;
using Vector3f = Vec<float, 3>;
using Vector2i = Vec<int, 2>;
using Vector3i = Vec<int, 3>;
using Array2i = Vec<int, 2>;
using Array2d = Vec<double, 2>;

template <typename T>
struct Mat3 {
  std::array<T, 9> m{};  // column-major
  T& operator()(int r, int c) { return m[size_t(c * 3 + r)]; }
  T operator()(int r, int c) const { return m[size_t(c * 3 + r)]; }
  static Mat3 Identity() { Mat3 a; a(0, 0) = a(1, 1) = a(2, 2) = T(1); return a; }
  static Mat3 Zero() { return Mat3(); }
  Mat3 operator*(const Mat3& o) const {
    Mat3 r;
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
        r(i, j) = (*this)(i, 0) * o(0, j) + ((*this)(i, 1) * o(1, j) + (*this)(i, 2) * o(2, j));
    return r;
  }
  Mat3 operator*(T s) const { Mat3 r; for (int i = 0; i < 9; ++i) r.m[size_t(i)] = m[size_t(i)] * s; return r; }
  Vec<T, 3> operator*(const Vec<T, 3>& v) const {
    return Vec<T, 3>((*this)(0, 0) * v[0] + (*this)(0, 1) * v[1] + (*this)(0, 2) * v[2],
                     (*this)(1, 0) * v[0] + (*this)(1, 1) * v[1] + (*this)(1, 2) * v[2],
                     (*this)(2, 0) * v[0] + (*this)(2, 1) * v[1] + (*this)(2, 2) * v[2]);
  }
  Mat3 transpose() const { Mat3 r; for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r(i, j) = (*this)(j, i); return r; }
  template <typename U> Mat3<U> cast() const { Mat3<U> r; for (int i = 0; i < 9; ++i) r.m[size_t(i)] = U(m[size_t(i)]); return r; }
};
using Matrix3f = Mat3<float>;
using Matrix3d = Mat3<double>;

struct AngleAxisd {
  double angle;
  Vector3d axis;
  AngleAxisd(double a, const Vector3d& ax) : angle(a), axis(ax) {}
  Matrix3d toRotationMatrix() const {
    const double c = std::cos(angle), s = std::sin(angle), t = 1.0 - c;
    const double x = axis[0], y = axis[1], z = axis[2];
    Matrix3d R;
    R(0, 0) = t * x * x + c;     R(0, 1) = t * x * y - s * z; R(0, 2) = t * x * z + s * y;
    R(1, 0) = t * x * y + s * z; R(1, 1) = t * y * y + c;     R(1, 2) = t * y * z - s * x;
    R(2, 0) = t * x * z - s * y; R(2, 1) = t * y * z + s * x; R(2, 2) = t * z * z + c;
    return R;
  }
};

// Rigid transform, 4x4 column-major like Eigen::Transform<double,3,Isometry>.
class Isometry3d {
 public:
  struct MatrixRef {
    const double* p;
    const double* data() const { return p; }
    double operator()(int r, int c) const { return p[c * 4 + r]; }
  };
  Isometry3d() { setIdentity(); }
  static Isometry3d Identity() { return Isometry3d(); }
  void setIdentity() { m_.fill(0.0); m_[0] = m_[5] = m_[10] = m_[15] = 1.0; }
  MatrixRef matrix() const { return MatrixRef{m_.data()}; }
  double* data() { return m_.data(); }
  const double* data() const { return m_.data(); }
  Vector3d translation() const { return Vector3d(m_[12], m_[13], m_[14]); }
  struct TranslationRef {  // T.translation().x() = 3.0;
    double* p;
    double& x() { return p[0]; }
    double& y() { return p[1]; }
    double& z() { return p[2]; }
    double& operator()(int i) { return p[i]; }
    template <int M = 2> Vec<double, 2> head() const { return Vec<double, 2>(p[0], p[1]); }
    operator Vector3d() const { return Vector3d(p[0], p[1], p[2]); }
    TranslationRef& operator=(const Vector3d& v) { p[0] = v[0]; p[1] = v[1]; p[2] = v[2]; return *this; }
  };
  TranslationRef translation() { return TranslationRef{m_.data() + 12}; }
  void setTranslation(const Vector3d& t) { m_[12] = t[0]; m_[13] = t[1]; m_[14] = t[2]; }
  Matrix3d linear() const {
    Matrix3d R;
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) R(r, c) = m_[size_t(c * 4 + r)];
    return R;
  }
  Matrix3d rotation() const { return linear(); }
  void setLinear(const Matrix3d& R) {
    for (int c = 0; c < 3; ++c) for (int r = 0; r < 3; ++r) m_[size_t(c * 4 + r)] = R(r, c);
  }
  // this = this * Translation(t)
  Isometry3d& translate(const Vector3d& t) {
    const Vector3d d = linear() * t;
    m_[12] += d[0]; m_[13] += d[1]; m_[14] += d[2];
    return *this;
  }
  // this = Translation(t) * this
  Isometry3d& pretranslate(const Vector3d& t) { m_[12] += t[0]; m_[13] += t[1]; m_[14] += t[2]; return *this; }
  Isometry3d& rotate(const AngleAxisd& aa) { setLinear(linear() * aa.toRotationMatrix()); return *this; }
  Isometry3d& rotate(const Matrix3d& R) { setLinear(linear() * R); return *this; }
  Isometry3d operator*(const Isometry3d& o) const {
    Isometry3d r;
    r.setLinear(linear() * o.linear());
    const Vector3d t = linear() * o.translation();
    r.setTranslation(Vector3d(t[0] + m_[12], t[1] + m_[13], t[2] + m_[14]));
    return r;
  }
  Vector3d operator*(const Vector3d& p) const {
    const Vector3d q = linear() * p;
    return Vector3d(q[0] + m_[12], q[1] + m_[13], q[2] + m_[14]);
  }

 private:
  std::array<double, 16> m_;
};

}  // namespace Eigen
#endif
