// grid_costmap.h - TEST INFRASTRUCTURE. Restatement of base_local_planner::CostmapModel::footprintCost of the ROS navigation stack on a
// plain uint8 grid (base_local_planner/src/costmap_model.cpp + include/base_local_planner/line_iterator.h + costmap_2d::Costmap2D::
// worldToMap; an un-vendored, unpinned dependency of the reference - package.xml <depend>base_local_planner</depend> - restated from the
// published noetic sources). Shared by the CPU oracle (teb_oracle.cpp) and by the stand-in CostmapModel the reference's own
// isTrajectoryFeasible is driven with in oracle/_ref (ref_shim/ref_driver.cpp): like the g2o stand-in, both sides of that pin use this
// one restatement, so the pin covers the reference's code (look-ahead rule, interpolation), not the navigation stack's.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdlib>

namespace gridcostmap {

struct Grid {
  const uint8_t* cells; int sx, sy; double res, ox, oy;
  // costmap_2d::Costmap2D::worldToMap
  bool world_to_map(double wx, double wy, unsigned int& mx, unsigned int& my) const {
    if (wx < ox || wy < oy) return false;
    mx = (unsigned int)(int)((wx - ox) / res);
    my = (unsigned int)(int)((wy - oy) / res);
    return mx < (unsigned int)sx && my < (unsigned int)sy;
  }
  // CostmapModel::pointCost: NO_INFORMATION 255 -> -2, LETHAL_OBSTACLE 254 -> -1
  double point_cost(int x, int y) const {
    const unsigned char cost = cells[(size_t)y * sx + x];
    if (cost == 255) return -2;
    if (cost == 254) return -1;
    return cost;
  }
  // CostmapModel::lineCost over base_local_planner::LineIterator (Bresenham)
  double line_cost(int x0, int x1, int y0, int y1) const {
    double line_cost = 0.0;
    const int deltax = std::abs(x1 - x0), deltay = std::abs(y1 - y0);
    int x = x0, y = y0;
    int xinc1, xinc2, yinc1, yinc2;
    if (x1 >= x0) { xinc1 = 1; xinc2 = 1; } else { xinc1 = -1; xinc2 = -1; }
    if (y1 >= y0) { yinc1 = 1; yinc2 = 1; } else { yinc1 = -1; yinc2 = -1; }
    int den, num, numadd, numpixels;
    if (deltax >= deltay) { xinc1 = 0; yinc2 = 0; den = deltax; num = deltax / 2; numadd = deltay; numpixels = deltax; }
    else { xinc2 = 0; yinc1 = 0; den = deltay; num = deltay / 2; numadd = deltax; numpixels = deltay; }
    for (int curpixel = 0; curpixel <= numpixels; ++curpixel) {
      const double pc = point_cost(x, y);
      if (pc < 0) return pc;
      if (line_cost < pc) line_cost = pc;
      num += numadd;
      if (num >= den) { num -= den; x += xinc1; y += yinc1; }
      x += xinc2; y += yinc2;
    }
    return line_cost;
  }
};

// WorldModel::footprintCost(x, y, theta, footprint_spec, ...) -> CostmapModel::footprintCost(position, oriented_footprint, ...):
// -1 lethal, -2 no information, -3 off the map, else the highest cell cost under the footprint outline
inline double footprint_cost(const Grid& g, double x, double y, double theta, int nf, const double* fx, const double* fy) {
  const double cos_th = std::cos(theta), sin_th = std::sin(theta);
  unsigned int cell_x, cell_y;
  if (!g.world_to_map(x, y, cell_x, cell_y)) return -3.0;
  if (nf < 3) {   // "we'll just assume a circular robot": the centre cell, where INSCRIBED_INFLATED_OBSTACLE 253 is lethal too
    const unsigned char cost = g.cells[(size_t)cell_y * g.sx + cell_x];
    if (cost == 255) return -2.0;
    if (cost == 254 || cost == 253) return -1.0;
    return cost;
  }
  double footprint_cost = 0.0;
  for (int i = 0; i < nf; ++i) {   // edges 0-1, 1-2, ..., then last-first
    const int j = (i + 1 < nf) ? i + 1 : 0;
    const double ax = x + (fx[i] * cos_th - fy[i] * sin_th), ay = y + (fx[i] * sin_th + fy[i] * cos_th);
    const double bx = x + (fx[j] * cos_th - fy[j] * sin_th), by = y + (fx[j] * sin_th + fy[j] * cos_th);
    unsigned int x0, y0, x1, y1;
    if (!g.world_to_map(ax, ay, x0, y0)) return -3.0;
    if (!g.world_to_map(bx, by, x1, y1)) return -3.0;
    const double lc = g.line_cost((int)x0, (int)x1, (int)y0, (int)y1);
    footprint_cost = std::max(lc, footprint_cost);
    if (lc < 0) return lc;
  }
  return footprint_cost;
}

}  // namespace gridcostmap
