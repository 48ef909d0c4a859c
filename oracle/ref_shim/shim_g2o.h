// shim_g2o.h — minimal stand-in for the libg2o classes the reference's vertex / edge headers derive from.
// TEST INFRASTRUCTURE (oracle/_ref build only). libg2o is an external, un-vendored dependency of the reference
// (package.xml:41); only the members the reference headers touch are provided. computeError() is the reference's own
// code; linearizeOplus() defaults restate g2o's central differences (delta = 1e-9) and are NOT what pins anything.
#pragma once
#include <cmath>
#include <iostream>
#include <stack>
#include <vector>

#include "shim_eigen.h"

namespace g2o {

inline double normalize_theta(double theta) {
  if (theta >= -M_PI && theta < M_PI) return theta;
  double multiplier = std::floor(theta / (2 * M_PI));
  theta = theta - multiplier * 2 * M_PI;
  if (theta >= M_PI) theta -= 2 * M_PI;
  if (theta < -M_PI) theta += 2 * M_PI;
  return theta;
}
template <typename T>
inline int sign(T x) { if (x > 0) return 1; else if (x < 0) return -1; else return 0; }
inline double average_angle(double theta1, double theta2) {
  double x = std::cos(theta1) + std::cos(theta2);
  double y = std::sin(theta1) + std::sin(theta2);
  if (x == 0 && y == 0) return 0;
  return std::atan2(y, x);
}

struct HyperGraph {
  struct Vertex {
    virtual ~Vertex() {}
  };
  struct HyperGraphElement {};
};
struct OptimizableGraph {
  struct Vertex : public HyperGraph::Vertex {
    bool _fixed = false;
    int _id = -1;
    bool fixed() const { return _fixed; }
    void setFixed(bool f) { _fixed = f; }
    void setId(int id) { _id = id; }
    int id() const { return _id; }
    virtual int dimension() const = 0;
    virtual void oplus(const double* v) = 0;
    virtual void push() = 0;
    virtual void pop() = 0;
  };
};

template <int D, typename T>
class BaseVertex : public OptimizableGraph::Vertex {
 public:
  static const int Dimension = D;
  typedef T EstimateType;
  const T& estimate() const { return _estimate; }
  void setEstimate(const T& e) { _estimate = e; }
  int dimension() const override { return D; }
  void oplus(const double* v) override { oplusImpl(v); }
  void push() override { _backup.push(_estimate); }
  void pop() override { _estimate = _backup.top(); _backup.pop(); }
  virtual void oplusImpl(const double* v) = 0;
  virtual void setToOriginImpl() = 0;
  virtual bool read(std::istream& is) = 0;
  virtual bool write(std::ostream& os) const = 0;
 protected:
  T _estimate;
  std::stack<T> _backup;
};

template <int D, typename E>
class BaseEdge {
 public:
  static const int Dimension = D;
  typedef E Measurement;
  typedef Eigen::Matrix<double, D, 1> ErrorVector;
  typedef Eigen::Matrix<double, D, D> InformationType;
  virtual ~BaseEdge() {}
  virtual void computeError() = 0;
  const ErrorVector& error() const { return _error; }
  ErrorVector& error() { return _error; }
  const InformationType& information() const { return _information; }
  InformationType& information() { return _information; }
  void setInformation(const InformationType& i) { _information = i; }
  const E& measurement() const { return _measurement; }
  virtual void setMeasurement(const E& m) { _measurement = m; }
  double chi2() const { ErrorVector t = _information * _error; return _error.dot(t); }
  void setVertex(size_t i, HyperGraph::Vertex* v) { _vertices[i] = v; }
  std::vector<HyperGraph::Vertex*>& vertices() { return _vertices; }
  virtual void resize(size_t n) { _vertices.resize(n, nullptr); }
 protected:
  E _measurement;
  ErrorVector _error;
  InformationType _information;
  std::vector<HyperGraph::Vertex*> _vertices;
};

template <int D, typename E, typename VertexXi>
class BaseUnaryEdge : public BaseEdge<D, E> {
 public:
  typedef Eigen::Matrix<double, D, VertexXi::Dimension> JacobianXiOplusType;
  BaseUnaryEdge() { this->_vertices.resize(1, nullptr); }
  virtual void linearizeOplus() {}
  const JacobianXiOplusType& jacobianOplusXi() const { return _jacobianOplusXi; }
 protected:
  JacobianXiOplusType _jacobianOplusXi;
};

template <int D, typename E, typename VertexXi, typename VertexXj>
class BaseBinaryEdge : public BaseEdge<D, E> {
 public:
  typedef Eigen::Matrix<double, D, VertexXi::Dimension> JacobianXiOplusType;
  typedef Eigen::Matrix<double, D, VertexXj::Dimension> JacobianXjOplusType;
  BaseBinaryEdge() { this->_vertices.resize(2, nullptr); }
  virtual void linearizeOplus() {}
  const JacobianXiOplusType& jacobianOplusXi() const { return _jacobianOplusXi; }
  const JacobianXjOplusType& jacobianOplusXj() const { return _jacobianOplusXj; }
 protected:
  JacobianXiOplusType _jacobianOplusXi;
  JacobianXjOplusType _jacobianOplusXj;
};

template <int D, typename E>
class BaseMultiEdge : public BaseEdge<D, E> {
 public:
  virtual void linearizeOplus() {}
};

}  // namespace g2o
