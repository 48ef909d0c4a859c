// shim_eigen.h — minimal stand-in for the subset of Eigen that the reference's optimiser-path headers use.
// TEST INFRASTRUCTURE (oracle/_ref build only). Fixed-size dense double matrices with the handful of members the
// reference calls; plain IEEE arithmetic in the same operation order as Eigen's coefficient-wise evaluation.
#pragma once
#include <cmath>
#include <cstddef>
#include <memory>
#include <vector>

#define EIGEN_MAKE_ALIGNED_OPERATOR_NEW

namespace Eigen {

template <typename T, int R, int C>
class Matrix {
 public:
  T d[R * C > 0 ? R * C : 1];
  Matrix() { for (int i = 0; i < R * C; ++i) d[i] = T(0); }
  Matrix(T a, T b) { static_assert(R * C == 2, "2-vector ctor"); d[0] = a; d[1] = b; }
  Matrix(T a, T b, T c) { static_assert(R * C == 3, "3-vector ctor"); d[0] = a; d[1] = b; d[2] = c; }
  static Matrix Zero() { return Matrix(); }
  static Matrix Identity() { Matrix m; for (int i = 0; i < (R < C ? R : C); ++i) m(i, i) = T(1); return m; }
  int rows() const { return R; }
  int cols() const { return C; }
  void resize(int, int) {}
  T& operator()(int i) { return d[i]; }
  const T& operator()(int i) const { return d[i]; }
  T& operator[](int i) { return d[i]; }
  const T& operator[](int i) const { return d[i]; }
  T& operator()(int i, int j) { return d[i + j * R]; }             // column-major like Eigen
  const T& operator()(int i, int j) const { return d[i + j * R]; }
  T& coeffRef(int i) { return d[i]; }
  const T& coeffRef(int i) const { return d[i]; }
  T& x() { return d[0]; }
  const T& x() const { return d[0]; }
  T& y() { return d[1]; }
  const T& y() const { return d[1]; }
  T& z() { return d[2]; }
  const T& z() const { return d[2]; }
  void setZero() { for (int i = 0; i < R * C; ++i) d[i] = T(0); }
  void setConstant(T v) { for (int i = 0; i < R * C; ++i) d[i] = v; }
  void fill(T v) { setConstant(v); }
  T squaredNorm() const { T s = T(0); for (int i = 0; i < R * C; ++i) s += d[i] * d[i]; return s; }
  T norm() const { return std::sqrt(squaredNorm()); }
  // first n coefficients of a vector as an aliasing 2-vector (only head(2) of a 3-vector is used: h_signature.h:302)
  Matrix<T, 2, 1>& head(int) { return *reinterpret_cast<Matrix<T, 2, 1>*>(d); }
  Matrix cross(const Matrix& o) const {   // Eigen: (a1 b2 - a2 b1, a2 b0 - a0 b2, a0 b1 - a1 b0)
    static_assert(R * C == 3, "cross of 3-vectors");
    Matrix m; m.d[0] = d[1] * o.d[2] - d[2] * o.d[1]; m.d[1] = d[2] * o.d[0] - d[0] * o.d[2]; m.d[2] = d[0] * o.d[1] - d[1] * o.d[0]; return m;
  }
  T dot(const Matrix& o) const { T s = T(0); for (int i = 0; i < R * C; ++i) s += d[i] * o.d[i]; return s; }
  Matrix normalized() const { T n = norm(); Matrix m(*this); if (n > T(0)) for (int i = 0; i < R * C; ++i) m.d[i] = d[i] / n; return m; }
  void normalize() { T n = norm(); if (n > T(0)) for (int i = 0; i < R * C; ++i) d[i] = d[i] / n; }
  bool isApprox(const Matrix& o, T prec = T(1e-12)) const {   // Eigen: ||a-b||^2 <= prec^2 * min(||a||^2, ||b||^2)
    Matrix df = *this - o;
    T m = squaredNorm() < o.squaredNorm() ? squaredNorm() : o.squaredNorm();
    return df.squaredNorm() <= prec * prec * m;
  }
  Matrix operator+(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] + o.d[i]; return m; }
  Matrix operator-(const Matrix& o) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] - o.d[i]; return m; }
  Matrix operator-() const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = -d[i]; return m; }
  Matrix operator*(T s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] * s; return m; }
  Matrix operator/(T s) const { Matrix m; for (int i = 0; i < R * C; ++i) m.d[i] = d[i] / s; return m; }
  Matrix& operator+=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d[i] += o.d[i]; return *this; }
  Matrix& operator-=(const Matrix& o) { for (int i = 0; i < R * C; ++i) d[i] -= o.d[i]; return *this; }
  Matrix& operator*=(T s) { for (int i = 0; i < R * C; ++i) d[i] *= s; return *this; }
  Matrix& operator/=(T s) { for (int i = 0; i < R * C; ++i) d[i] /= s; return *this; }
  template <int C2>
  Matrix<T, R, C2> operator*(const Matrix<T, C, C2>& o) const {
    Matrix<T, R, C2> m;
    for (int i = 0; i < R; ++i) for (int j = 0; j < C2; ++j) { T s = T(0); for (int k = 0; k < C; ++k) s += (*this)(i, k) * o(k, j); m(i, j) = s; }
    return m;
  }
};
template <typename T, int R, int C>
inline Matrix<T, R, C> operator*(T s, const Matrix<T, R, C>& m) { return m * s; }
template <typename T, int R, int C>
inline Matrix<T, R, C> operator*(int s, const Matrix<T, R, C>& m) { return m * T(s); }

typedef Matrix<double, 2, 1> Vector2d;
typedef Matrix<double, 3, 1> Vector3d;
typedef Matrix<double, 2, 2> Matrix2d;
typedef Matrix<double, 3, 3> Matrix3d;

// Eigen::Ref: a const Ref may bind to temporaries (it then owns a copy); a mutable Ref aliases its target.
template <typename V>
class Ref;
template <typename T, int R, int C>
class Ref<const Matrix<T, R, C>> : public Matrix<T, R, C> {
 public:
  Ref(const Matrix<T, R, C>& v) : Matrix<T, R, C>(v) {}
};
template <typename T, int R, int C>
class Ref<Matrix<T, R, C>> {
  Matrix<T, R, C>* p_;
 public:
  Ref(Matrix<T, R, C>& v) : p_(&v) {}
  Ref& operator=(const Matrix<T, R, C>& v) { *p_ = v; return *this; }
  operator Matrix<T, R, C>&() { return *p_; }
  T& x() { return p_->x(); }
  T& y() { return p_->y(); }
  T& coeffRef(int i) { return p_->coeffRef(i); }
};

template <typename T>
using aligned_allocator = std::allocator<T>;

class Rotation2Dd {
  double a_;
 public:
  explicit Rotation2Dd(double a) : a_(a) {}
  Vector2d operator*(const Vector2d& v) const { return Vector2d(std::cos(a_) * v.x() - std::sin(a_) * v.y(), std::sin(a_) * v.x() + std::cos(a_) * v.y()); }
  Matrix2d toRotationMatrix() const { Matrix2d m; m(0, 0) = std::cos(a_); m(0, 1) = -std::sin(a_); m(1, 0) = std::sin(a_); m(1, 1) = std::cos(a_); return m; }
};

template <typename T> class Rotation2D;
template <> class Rotation2D<double> : public Rotation2Dd { public: explicit Rotation2D(double a) : Rotation2Dd(a) {} };

}  // namespace Eigen
