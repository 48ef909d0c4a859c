// shim_g2o.h — minimal stand-in for the libg2o classes the reference derives from / drives.
// TEST INFRASTRUCTURE (oracle/_ref build only). libg2o is an external, un-vendored dependency of the reference
// (package.xml:41). Only what the reference touches is provided:
//   * vertex / edge base classes with g2o's member names, so the reference's OWN computeError() / linearizeOplus()
//     / oplusImpl() code compiles unchanged;
//   * the default numeric linearizeOplus (central differences, delta = 1e-9) of Base{Unary,Binary,Multi}Edge;
//   * a SparseOptimizer that RECORDS the graph the reference's buildGraph() creates and runs a restatement of
//     OptimizationAlgorithmLevenberg on it (banded normal equations + Cholesky) — SURVEY.md Appendix B.
// The restated parts pin nothing by themselves; they let the reference's real graph construction, weights, edge
// order, cost evaluation and outer loop (src/optimal_planner.cpp) run end-to-end for comparison with the oracle.
#pragma once
#include <algorithm>
#include <cmath>
#include <iostream>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <stack>
#include <string>
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
  struct Edge;
  typedef std::set<Edge*> EdgeSet;
  struct HyperGraphElement { virtual ~HyperGraphElement() {} };
  struct Vertex : public HyperGraphElement {
    EdgeSet _edges;
    EdgeSet& edges() { return _edges; }
  };
  struct Edge : public HyperGraphElement {
    long long _internalId = -1;
  };
  typedef std::map<int, Vertex*> VertexIDMap;
};

struct OptimizableGraph : public HyperGraph {
  struct Vertex : public HyperGraph::Vertex {
    bool _fixed = false;
    int _id = -1;
    int _hessianIndex = -1;
    bool fixed() const { return _fixed; }
    void setFixed(bool f) { _fixed = f; }
    void setId(int id) { _id = id; }
    int id() const { return _id; }
    virtual int dimension() const = 0;
    virtual void oplus(const double* v) = 0;
    virtual void push() = 0;
    virtual void pop() = 0;
    virtual void discardTop() = 0;
  };
  struct Edge : public HyperGraph::Edge {
    virtual void computeError() = 0;
    virtual void linearizeOplus() = 0;
    virtual double chi2() const = 0;
    virtual int dimension() const = 0;
    virtual double errorAt(int k) const = 0;
    virtual double infoAt(int r, int c) const = 0;
    virtual int nVertices() const = 0;
    virtual OptimizableGraph::Vertex* vertexAt(int i) const = 0;
    virtual double jacAt(int vi, int r, int c) const = 0;
    bool allVerticesFixed() const {
      for (int i = 0; i < nVertices(); ++i) if (!vertexAt(i)->fixed()) return false;
      return true;
    }
  };
  typedef std::vector<Edge*> EdgeContainer;
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
  void discardTop() override { _backup.pop(); }
  virtual void oplusImpl(const double* v) = 0;
  virtual void setToOriginImpl() = 0;
  virtual bool read(std::istream& is) = 0;
  virtual bool write(std::ostream& os) const = 0;
 protected:
  T _estimate;
  std::stack<T> _backup;
};

template <int D, typename E>
class BaseEdge : public OptimizableGraph::Edge {
 public:
  static const int Dimension = D;
  typedef E Measurement;
  typedef Eigen::Matrix<double, D, 1> ErrorVector;
  typedef Eigen::Matrix<double, D, D> InformationType;
  const ErrorVector& error() const { return _error; }
  ErrorVector& error() { return _error; }
  const InformationType& information() const { return _information; }
  InformationType& information() { return _information; }
  void setInformation(const InformationType& i) { _information = i; }
  const E& measurement() const { return _measurement; }
  virtual void setMeasurement(const E& m) { _measurement = m; }
  double chi2() const override { ErrorVector t = _information * _error; return _error.dot(t); }   // _error.dot(information()*_error)
  void setVertex(size_t i, HyperGraph::Vertex* v) { _vertices[i] = v; }
  std::vector<HyperGraph::Vertex*>& vertices() { return _vertices; }
  virtual void resize(size_t n) { _vertices.resize(n, nullptr); }
  int dimension() const override { return D; }
  double errorAt(int k) const override { return _error[k]; }
  double infoAt(int r, int c) const override { return _information(r, c); }
  int nVertices() const override { return (int)_vertices.size(); }
  OptimizableGraph::Vertex* vertexAt(int i) const override { return static_cast<OptimizableGraph::Vertex*>(_vertices[i]); }
 protected:
  E _measurement;
  ErrorVector _error;
  InformationType _information;
  std::vector<HyperGraph::Vertex*> _vertices;
};

// numeric Jacobian of one vertex block, exactly g2o's scheme: push; oplus(+delta e_d); computeError; pop; push;
// oplus(-delta e_d); computeError; pop; column = scalar * (e+ - e-), delta = 1e-9, scalar = 1/(2 delta)
template <class EdgeT, class Store>
inline void numeric_block(EdgeT* e, OptimizableGraph::Vertex* vi, Store store) {
  const double delta = 1e-9;
  const double scalar = 1.0 / (2 * delta);
  const int vd = vi->dimension();
  double add[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int d = 0; d < vd; ++d) {
    vi->push();
    add[d] = delta;
    vi->oplus(add);
    e->computeError();
    typename EdgeT::ErrorVector e1 = e->error();
    vi->pop();
    vi->push();
    add[d] = -delta;
    vi->oplus(add);
    e->computeError();
    vi->pop();
    add[d] = 0.0;
    typename EdgeT::ErrorVector diff = e1 - e->error();
    for (int k = 0; k < EdgeT::Dimension; ++k) store(k, d, scalar * diff[k]);
  }
}

template <int D, typename E, typename VertexXi>
class BaseUnaryEdge : public BaseEdge<D, E> {
 public:
  typedef Eigen::Matrix<double, D, VertexXi::Dimension> JacobianXiOplusType;
  BaseUnaryEdge() { this->_vertices.resize(1, nullptr); }
  void linearizeOplus() override {
    OptimizableGraph::Vertex* vi = this->vertexAt(0);
    if (vi->fixed()) return;
    typename BaseEdge<D, E>::ErrorVector before = this->_error;
    numeric_block(this, vi, [this](int k, int d, double v) { _jacobianOplusXi(k, d) = v; });
    this->_error = before;
  }
  const JacobianXiOplusType& jacobianOplusXi() const { return _jacobianOplusXi; }
  double jacAt(int, int r, int c) const override { return _jacobianOplusXi(r, c); }
 protected:
  JacobianXiOplusType _jacobianOplusXi;
};

template <int D, typename E, typename VertexXi, typename VertexXj>
class BaseBinaryEdge : public BaseEdge<D, E> {
 public:
  typedef Eigen::Matrix<double, D, VertexXi::Dimension> JacobianXiOplusType;
  typedef Eigen::Matrix<double, D, VertexXj::Dimension> JacobianXjOplusType;
  BaseBinaryEdge() { this->_vertices.resize(2, nullptr); }
  void linearizeOplus() override {
    OptimizableGraph::Vertex* vi = this->vertexAt(0);
    OptimizableGraph::Vertex* vj = this->vertexAt(1);
    if (vi->fixed() && vj->fixed()) return;
    typename BaseEdge<D, E>::ErrorVector before = this->_error;
    if (!vi->fixed()) numeric_block(this, vi, [this](int k, int d, double v) { _jacobianOplusXi(k, d) = v; });
    if (!vj->fixed()) numeric_block(this, vj, [this](int k, int d, double v) { _jacobianOplusXj(k, d) = v; });
    this->_error = before;
  }
  const JacobianXiOplusType& jacobianOplusXi() const { return _jacobianOplusXi; }
  const JacobianXjOplusType& jacobianOplusXj() const { return _jacobianOplusXj; }
  double jacAt(int vi, int r, int c) const override { return vi == 0 ? _jacobianOplusXi(r, c) : _jacobianOplusXj(r, c); }
 protected:
  JacobianXiOplusType _jacobianOplusXi;
  JacobianXjOplusType _jacobianOplusXj;
};

template <int D, typename E>
class BaseMultiEdge : public BaseEdge<D, E> {
 public:
  void resize(size_t n) override { BaseEdge<D, E>::resize(n); _jac.assign(n, std::vector<double>(D * 8, 0.0)); }
  void linearizeOplus() override {
    typename BaseEdge<D, E>::ErrorVector before = this->_error;
    for (int i = 0; i < this->nVertices(); ++i) {
      OptimizableGraph::Vertex* vi = this->vertexAt(i);
      if (vi->fixed()) continue;
      std::vector<double>& J = _jac[i];
      numeric_block(this, vi, [&J](int k, int d, double v) { J[k * 8 + d] = v; });
    }
    this->_error = before;
  }
  double jacAt(int vi, int r, int c) const override { return _jac[vi][r * 8 + c]; }
 protected:
  std::vector<std::vector<double>> _jac;
};

// ---- factory / solver scaffolding used by TebOptimalPlanner::registerG2OTypes / initOptimizer -------------------------
struct AbstractHyperGraphElementCreator { virtual ~AbstractHyperGraphElementCreator() {} };
template <typename T> struct HyperGraphElementCreator : public AbstractHyperGraphElementCreator {};
class Factory {
 public:
  static Factory* instance() { static Factory f; return &f; }
  void registerType(const std::string&, AbstractHyperGraphElementCreator* c) { delete c; }
};
template <int P, int L> struct BlockSolverTraits { typedef int PoseMatrixType; };
template <typename MatrixT> class LinearSolverCSparse { public: void setBlockOrdering(bool) {} };
template <typename MatrixT> class LinearSolverCholmod { public: void setBlockOrdering(bool) {} };
template <typename Traits>
class BlockSolver {
 public:
  typedef typename Traits::PoseMatrixType PoseMatrixType;
  template <typename LS> explicit BlockSolver(std::unique_ptr<LS>) {}
};
class OptimizationAlgorithm { public: virtual ~OptimizationAlgorithm() {} };
class OptimizationAlgorithmLevenberg : public OptimizationAlgorithm {
 public:
  template <typename BS> explicit OptimizationAlgorithmLevenberg(std::unique_ptr<BS>) {}
};
class OptimizationAlgorithmGaussNewton : public OptimizationAlgorithm {
 public:
  template <typename BS> explicit OptimizationAlgorithmGaussNewton(std::unique_ptr<BS>) {}
};
struct G2OBatchStatistics { int iteration = 0; double chi2 = 0; };
typedef std::vector<G2OBatchStatistics> BatchStatisticsContainer;

// ---- SparseOptimizer: records the graph, then a restated Levenberg-Marquardt -----------------
class SparseOptimizer : public OptimizableGraph {
 public:
  ~SparseOptimizer() { clear(); delete _algorithm; }
  VertexIDMap& vertices() { return _vertices; }
  EdgeSet& edges() { return _edges; }
  bool addVertex(OptimizableGraph::Vertex* v) { _vertices[v->id()] = v; return true; }
  bool addEdge(OptimizableGraph::Edge* e) {
    e->_internalId = _nextEdgeId++;
    _edges.insert(e);
    _edgeOrder.push_back(e);
    for (int i = 0; i < e->nVertices(); ++i) e->vertexAt(i)->edges().insert(e);
    return true;
  }
  void clear() {   // g2o deletes the graph elements it owns; the reference empties vertices() first so the TEB survives
    for (auto& kv : _vertices) delete kv.second;
    _vertices.clear();
    for (OptimizableGraph::Edge* e : _edgeOrder) delete e;
    _edges.clear(); _edgeOrder.clear(); _active.clear(); _index.clear();
  }
  void setAlgorithm(OptimizationAlgorithm* a) { delete _algorithm; _algorithm = a; }
  void initMultiThreading() {}
  void setVerbose(bool) {}
  void setComputeBatchStatistics(bool b) { _stats = b; }
  const BatchStatisticsContainer& batchStatistics() const { return _batch; }
  void computeInitialGuess() {}
  const EdgeContainer& activeEdges() const { return _active; }
  bool initializeOptimization(int = 0) {
    _active.clear(); _index.clear();
    for (OptimizableGraph::Edge* e : _edgeOrder) if (!e->allVerticesFixed()) _active.push_back(e);   // insertion (= internal id) order
    int off = 0;
    for (auto& kv : _vertices) {   // std::map: ascending vertex id
      OptimizableGraph::Vertex* v = static_cast<OptimizableGraph::Vertex*>(kv.second);
      if (v->fixed()) { v->_hessianIndex = -1; continue; }
      v->_hessianIndex = off; off += v->dimension(); _index.push_back(v);
    }
    _N = off;
    _KD = 0;   // half bandwidth of the normal matrix in this variable order
    for (OptimizableGraph::Edge* e : _active) {
      int lo = 1 << 30, hi = -1;
      for (int i = 0; i < e->nVertices(); ++i) {
        OptimizableGraph::Vertex* v = e->vertexAt(i);
        if (v->fixed()) continue;
        lo = std::min(lo, v->_hessianIndex); hi = std::max(hi, v->_hessianIndex + v->dimension() - 1);
      }
      if (hi >= 0) _KD = std::max(_KD, hi - lo);
    }
    return true;
  }
  void computeActiveErrors() { for (OptimizableGraph::Edge* e : _active) e->computeError(); }
  double activeChi2() const { double c = 0; for (OptimizableGraph::Edge* e : _active) c += e->chi2(); return c; }

  static long& iterationCounter() { static thread_local long c = 0; return c; }   // LM iterations run by this thread (bench accounting)
  // opt-in trace of the LM loop run by this thread (ref_set_trace in ref_driver.cpp): one row per LM iteration = {chi2 after it, lambda
  // after it, damping trials, pose count (from the system size N = 4 n - 7)} - what g2o prints per iteration in verbose mode
  struct LmTrace { double* buf = nullptr; int cap = 0; int32_t* rows = nullptr; };
  static LmTrace& lmTrace() { static thread_local LmTrace t; return t; }
  int optimize(int iterations) {
    if (_index.empty()) return -1;
    _batch.clear();
    if (_stats) _batch.resize(iterations);
    int cj = 0;
    bool ok = true;
    for (int i = 0; i < iterations && ok; ++i) {
      ok = solveLM(i);
      if (_stats) { computeActiveErrors(); _batch[i].iteration = i; _batch[i].chi2 = activeChi2(); }
      ++cj;
    }
    iterationCounter() += cj;
    return cj;
  }

 private:
  bool solveLM(int iteration) {
    const int N = _N;
    computeActiveErrors();
    double currentChi = activeChi2();
    double tempChi = currentChi;
    const int W = _KD + 1;   // band storage: H(ia, ic) at [ia * W + (ia - ic)], ic <= ia (the zeros outside the band never enter the arithmetic)
    std::vector<double> H((size_t)N * W, 0.0), b(N, 0.0), x(N, 0.0);
    for (OptimizableGraph::Edge* e : _active) {
      e->linearizeOplus();
      const int D = e->dimension();
      for (int i = 0; i < e->nVertices(); ++i) {
        OptimizableGraph::Vertex* vi = e->vertexAt(i);
        if (vi->fixed()) continue;
        for (int a = 0; a < vi->dimension(); ++a) {
          const int ia = vi->_hessianIndex + a;
          double bb = 0;
          for (int k = 0; k < D; ++k) bb += e->jacAt(i, k, a) * (-(e->infoAt(k, k) * e->errorAt(k)));
          b[ia] += bb;
          for (int j = 0; j < e->nVertices(); ++j) {
            OptimizableGraph::Vertex* vj = e->vertexAt(j);
            if (vj->fixed()) continue;
            for (int c = 0; c < vj->dimension(); ++c) {
              const int ic = vj->_hessianIndex + c;
              if (ic > ia) continue;
              double h = 0;
              for (int k = 0; k < D; ++k) h += (e->jacAt(i, k, a) * e->infoAt(k, k)) * e->jacAt(j, k, c);
              H[(size_t)ia * W + (ia - ic)] += h;
            }
          }
        }
      }
    }
    if (iteration == 0) {
      double maxDiagonal = 0;
      for (int k = 0; k < N; ++k) maxDiagonal = std::max(std::fabs(H[(size_t)k * W]), maxDiagonal);
      _lambda = 1e-5 * maxDiagonal;
      _ni = 2;
    }
    double rho = 0;
    int qmax = 0;
    do {
      for (OptimizableGraph::Vertex* v : _index) v->push();
      bool ok2 = cholSolve(H, b, x, N, _KD, _lambda);
      if (!ok2) x = b;
      for (OptimizableGraph::Vertex* v : _index) v->oplus(&x[v->_hessianIndex]);
      computeActiveErrors();
      tempChi = activeChi2();
      if (!ok2) tempChi = std::numeric_limits<double>::max();
      rho = (currentChi - tempChi);
      double scale = 0;
      for (int j = 0; j < N; ++j) scale += x[j] * (_lambda * x[j] + b[j]);
      scale += 1e-3;
      rho /= scale;
      if (rho > 0 && std::isfinite(tempChi)) {
        double alpha = 1. - std::pow((2 * rho - 1), 3);
        alpha = std::min(alpha, 2. / 3.);
        double scaleFactor = std::max(1. / 3., alpha);
        _lambda *= scaleFactor;
        _ni = 2;
        currentChi = tempChi;
        for (OptimizableGraph::Vertex* v : _index) v->discardTop();
      } else {
        _lambda *= _ni;
        _ni *= 2;
        for (OptimizableGraph::Vertex* v : _index) v->pop();
        if (!std::isfinite(_lambda)) break;
      }
      qmax++;
    } while (rho < 0 && qmax < 10);
    LmTrace& tr = lmTrace();
    if (tr.buf && tr.rows && *tr.rows < tr.cap) {
      double* row = tr.buf + (size_t)(*tr.rows) * 4;
      row[0] = currentChi; row[1] = _lambda; row[2] = (double)qmax; row[3] = (double)((N + 7) / 4);
      ++*tr.rows;
    }
    if (qmax == 10 || rho == 0 || !std::isfinite(_lambda)) return false;
    return true;
  }
  // (H + lambda I) x = b by banded Cholesky; fails iff a pivot is <= 0, the condition under which CSparse's cs_chol gives up
  static bool cholSolve(const std::vector<double>& H, const std::vector<double>& b, std::vector<double>& x, int N, int KD, double lambda) {
    const int W = KD + 1;
    std::vector<double> L((size_t)N * W, 0.0);
    for (int j = 0; j < N; ++j) {
      const int c0 = std::max(0, j - KD);
      for (int c = c0; c <= j; ++c) {
        double sum = H[(size_t)j * W + (j - c)];
        if (c == j) sum += lambda;
        for (int k = std::max(c0, c - KD); k < c; ++k) sum -= L[(size_t)j * W + (j - k)] * L[(size_t)c * W + (c - k)];
        if (c == j) { if (sum <= 0) return false; L[(size_t)j * W] = std::sqrt(sum); }
        else L[(size_t)j * W + (j - c)] = sum / L[(size_t)c * W];
      }
    }
    for (int i = 0; i < N; ++i) { double s = b[i]; for (int k = std::max(0, i - KD); k < i; ++k) s -= L[(size_t)i * W + (i - k)] * x[k]; x[i] = s / L[(size_t)i * W]; }
    for (int i = N - 1; i >= 0; --i) { double s = x[i]; for (int k = i + 1; k <= std::min(N - 1, i + KD); ++k) s -= L[(size_t)k * W + (k - i)] * x[k]; x[i] = s / L[(size_t)i * W]; }
    return true;
  }

  VertexIDMap _vertices;
  EdgeSet _edges;
  std::vector<OptimizableGraph::Edge*> _edgeOrder;
  EdgeContainer _active;
  std::vector<OptimizableGraph::Vertex*> _index;
  OptimizationAlgorithm* _algorithm = nullptr;
  long long _nextEdgeId = 0;
  int _N = 0, _KD = 0;
  bool _stats = false;
  BatchStatisticsContainer _batch;
  double _lambda = 0, _ni = 2;
};

}  // namespace g2o
