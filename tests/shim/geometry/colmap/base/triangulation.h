// TEST SHIM of the one function the adapter uses from src/geometry/colmap/base/triangulation.h
#pragma once
#include <vector>
#include "base/map.h"
namespace colmap {
std::vector<double> CalculateTriangulationAngles(const xrsfm::vector3 &c1, const xrsfm::vector3 &c2,
                                                 const std::vector<xrsfm::vector3> &points3D);
}
