// Stand-ins for the Boost pieces HomotopyClassPlanner / graph_search.cpp use (Boost is absent from this image).
// TEST INFRASTRUCTURE (oracle/_ref build only). Semantics restated from the Boost 1.7x documentation:
//  * adjacency_list<listS, vecS, directedS, VP>: vertex descriptors are indices 0..N-1 in insertion order, the out-edge
//    container of a vertex is a list in insertion order (adjacent_vertices iterates it front to back, parallel edges allowed);
//  * mt19937: the 32-bit Mersenne twister, default seed 5489 (std::mt19937 is the same engine);
//  * uniform_real_distribution<double>(a, b) on a 32-bit engine (boost/random/uniform_real_distribution.hpp, generate_uniform_real):
//    loop { r = double(eng() - eng.min()) / (double(eng.max() - eng.min()) + 1) * (b - a) + a; if (r < b) return r; } -
//    ONE engine draw per value (std::uniform_real_distribution<double> draws two).
#ifndef TEB_REF_SHIM_BOOST_GRAPH_H_
#define TEB_REF_SHIM_BOOST_GRAPH_H_
#include <functional>
#include <list>
#include <random>
#include <thread>
#include <tuple>
#include <utility>
#include <vector>
#include "shim_ros.h"

using namespace std::placeholders;   // boost's _1, _2 live in the global namespace

namespace boost {
struct listS {}; struct vecS {}; struct directedS {}; struct no_property {};
template <class OutEdgeS, class VertexS, class DirS, class VP, class EP = no_property>
class adjacency_list {
 public:
  typedef std::size_t vertex_descriptor;
  typedef std::pair<std::size_t, std::size_t> edge_descriptor;
  typedef std::list<vertex_descriptor>::const_iterator adjacency_iterator;
  class vertex_iterator {   // random-access counting iterator
    std::size_t i_ = 0;
   public:
    typedef std::random_access_iterator_tag iterator_category;
    typedef std::size_t value_type; typedef std::ptrdiff_t difference_type; typedef const std::size_t* pointer; typedef std::size_t reference;
    vertex_iterator() {}
    explicit vertex_iterator(std::size_t i) : i_(i) {}
    std::size_t operator*() const { return i_; }
    vertex_iterator& operator++() { ++i_; return *this; }
    vertex_iterator operator++(int) { vertex_iterator t = *this; ++i_; return t; }
    vertex_iterator& operator--() { --i_; return *this; }
    vertex_iterator& operator+=(std::ptrdiff_t d) { i_ += d; return *this; }
    vertex_iterator& operator-=(std::ptrdiff_t d) { i_ -= d; return *this; }
    vertex_iterator operator-(std::ptrdiff_t d) const { return vertex_iterator(i_ - d); }
    vertex_iterator operator+(std::ptrdiff_t d) const { return vertex_iterator(i_ + d); }
    std::ptrdiff_t operator-(const vertex_iterator& o) const { return (std::ptrdiff_t)i_ - (std::ptrdiff_t)o.i_; }
    bool operator==(const vertex_iterator& o) const { return i_ == o.i_; }
    bool operator!=(const vertex_iterator& o) const { return i_ != o.i_; }
  };
  typedef void edge_iterator;
  std::vector<VP> props;
  std::vector<std::list<vertex_descriptor>> out;
  VP& operator[](vertex_descriptor v) { return props[v]; }
  const VP& operator[](vertex_descriptor v) const { return props[v]; }
  void clear() { props.clear(); out.clear(); }
};
template <class G> struct graph_traits {
  typedef typename G::vertex_descriptor vertex_descriptor;
  typedef typename G::edge_descriptor edge_descriptor;
  typedef typename G::vertex_iterator vertex_iterator;
  typedef typename G::edge_iterator edge_iterator;
  typedef typename G::adjacency_iterator adjacency_iterator;
};
template <class A, class B, class C, class VP, class EP>
std::size_t add_vertex(adjacency_list<A, B, C, VP, EP>& g) { g.props.emplace_back(); g.out.emplace_back(); return g.props.size() - 1; }
template <class A, class B, class C, class VP, class EP>
std::pair<std::pair<std::size_t, std::size_t>, bool> add_edge(std::size_t u, std::size_t v, adjacency_list<A, B, C, VP, EP>& g) {
  g.out[u].push_back(v); return {{u, v}, true};
}
template <class A, class B, class C, class VP, class EP>
std::size_t num_vertices(const adjacency_list<A, B, C, VP, EP>& g) { return g.props.size(); }
template <class A, class B, class C, class VP, class EP>
std::pair<typename adjacency_list<A, B, C, VP, EP>::vertex_iterator, typename adjacency_list<A, B, C, VP, EP>::vertex_iterator>
vertices(const adjacency_list<A, B, C, VP, EP>& g) {
  typedef typename adjacency_list<A, B, C, VP, EP>::vertex_iterator It;
  return {It(0), It(g.props.size())};
}
template <class A, class B, class C, class VP, class EP>
std::pair<typename adjacency_list<A, B, C, VP, EP>::adjacency_iterator, typename adjacency_list<A, B, C, VP, EP>::adjacency_iterator>
adjacent_vertices(std::size_t v, const adjacency_list<A, B, C, VP, EP>& g) { return {g.out[v].begin(), g.out[v].end()}; }

using std::tie;
using std::cref;
using std::ref;
template <class F, class... A> auto bind(F&& f, A&&... a) -> decltype(std::bind(std::forward<F>(f), std::forward<A>(a)...)) {
  return std::bind(std::forward<F>(f), std::forward<A>(a)...);
}

namespace random {
typedef std::mt19937 mt19937;
template <class T = double>
class uniform_real_distribution {
  T a_, b_;
 public:
  uniform_real_distribution(T a = 0, T b = 1) : a_(a), b_(b) {}
  template <class Eng> T operator()(Eng& eng) const {
    for (;;) {
      const T numerator = static_cast<T>(eng() - (Eng::min)());
      const T divisor = static_cast<T>((Eng::max)() - (Eng::min)()) + 1;
      const T result = numerator / divisor * (b_ - a_) + a_;
      if (result < b_) return result;
    }
  }
};
}  // namespace random
using random::mt19937;

// boost::thread_group as used by HomotopyClassPlanner::optimizeAllTEBs (enable_multithreading)
class thread_group {
  std::vector<std::thread> t_;
 public:
  template <class F> void create_thread(F f) { t_.emplace_back(f); }
  void join_all() { for (auto& t : t_) t.join(); t_.clear(); }
  ~thread_group() { join_all(); }
};
namespace this_thread { struct disable_interruption {}; }
}  // namespace boost
#endif
