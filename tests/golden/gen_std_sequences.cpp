// Generates the libstdc++-specific random sequences the reference's P2 tests draw
// (fastdem/tests/test_quantile_estimation.cpp:69-119): std::mt19937(42) through
// std::uniform_real_distribution<float>(0,10) x100 and std::normal_distribution<float>(5,1) x1000.
// Output: raw little-endian float32.  Build+run: g++ -O2 gen_std_sequences.cpp -o /tmp/gen && /tmp/gen <dir>
#include <cstdio>
#include <random>
#include <string>
#include <vector>

static void dump(const std::string& path, const std::vector<float>& v) {
  FILE* f = std::fopen(path.c_str(), "wb");
  std::fwrite(v.data(), sizeof(float), v.size(), f);
  std::fclose(f);
}

int main(int argc, char** argv) {
  const std::string dir = argc > 1 ? argv[1] : ".";
  {
    std::mt19937 gen(42);
    std::uniform_real_distribution<float> dist(0.0f, 10.0f);
    std::vector<float> v(100);
    for (auto& x : v) x = dist(gen);
    dump(dir + "/mt19937_42_uniform_0_10_x100.f32", v);
  }
  {
    std::mt19937 gen(42);
    std::normal_distribution<float> dist(5.0f, 1.0f);
    std::vector<float> v(1000);
    for (auto& x : v) x = dist(gen);
    dump(dir + "/mt19937_42_normal_5_1_x1000.f32", v);
  }
  return 0;
}
