// fastdem/point_types.hpp (fastdem/include/fastdem/point_types.hpp)
#pragma once
#include <nanopcl/core.hpp>
namespace fastdem {
using PointCloud = nanopcl::PointCloud;
using Point = nanopcl::Point;
using Color = nanopcl::Color;
}  // namespace fastdem
