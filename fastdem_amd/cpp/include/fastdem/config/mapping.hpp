// fastdem/config/mapping.hpp (fastdem/include/fastdem/config/mapping.hpp:10-48) — same names, same defaults.
#pragma once
namespace fastdem {
enum class MappingMode { LOCAL, GLOBAL };
enum class EstimationType { Kalman, P2Quantile };
namespace config {
struct Kalman {
  float min_variance = 0.0001f;
  float max_variance = 0.01f;
  float process_noise = 0.0f;
};
struct P2Quantile {
  float dn0 = 0.01f, dn1 = 0.16f, dn2 = 0.50f, dn3 = 0.84f, dn4 = 0.99f;
  int elevation_marker = 3;
  float max_sample_count = 0.0f;
};
struct Mapping {
  MappingMode mode = MappingMode::LOCAL;
  EstimationType estimation_type = EstimationType::Kalman;
  Kalman kalman;
  P2Quantile p2;
};
}  // namespace config
}  // namespace fastdem
