// mini_test.hpp — a few gtest-shaped macros (GoogleTest is not available offline).
#pragma once
#include <cmath>
#include <cstdio>
#include <cstring>
#include <functional>
#include <string>
#include <vector>

namespace mini {
struct Case { std::string name; std::function<void()> fn; };
inline std::vector<Case>& cases() { static std::vector<Case> c; return c; }
inline int& failures() { static int f = 0; return f; }
struct Reg { Reg(const char* n, std::function<void()> f) { cases().push_back({n, std::move(f)}); } };
inline bool float_eq(float a, float b) {  // EXPECT_FLOAT_EQ: within 4 ULPs
  if (a == b) return true;
  if (std::isnan(a) || std::isnan(b)) return false;
  int ia, ib; std::memcpy(&ia, &a, 4); std::memcpy(&ib, &b, 4);
  if ((ia < 0) != (ib < 0)) return false;
  return std::abs(ia - ib) <= 4;
}
inline int run(const char* filter) {
  int ran = 0;
  for (auto& c : cases()) {
    if (filter && c.name.find(filter) == std::string::npos) continue;
    const int before = failures();
    try { c.fn(); } catch (const std::exception& e) {
      std::printf("  EXCEPTION in %s: %s\n", c.name.c_str(), e.what()); ++failures();
    }
    std::printf("[%s] %s\n", failures() == before ? "  OK  " : "FAILED", c.name.c_str());
    ++ran;
  }
  std::printf("%d tests, %d failures\n", ran, failures());
  return failures() ? 1 : 0;
}
}  // namespace mini

#define TEST(suite, name) \
  static void suite##_##name(); \
  static mini::Reg reg_##suite##_##name(#suite "." #name, suite##_##name); \
  static void suite##_##name()
#define FAIL_MSG(msg) do { std::printf("  %s:%d: %s\n", __FILE__, __LINE__, msg); ++mini::failures(); } while (0)
#define EXPECT_TRUE(x) do { if (!(x)) FAIL_MSG("EXPECT_TRUE(" #x ")"); } while (0)
#define EXPECT_FALSE(x) do { if (x) FAIL_MSG("EXPECT_FALSE(" #x ")"); } while (0)
#define ASSERT_TRUE(x) do { if (!(x)) { FAIL_MSG("ASSERT_TRUE(" #x ")"); return; } } while (0)
#define ASSERT_EQ(a, b) do { if (!((a) == (b))) { FAIL_MSG("ASSERT_EQ(" #a ", " #b ")"); return; } } while (0)
#define EXPECT_EQ(a, b) do { if (!((a) == (b))) FAIL_MSG("EXPECT_EQ(" #a ", " #b ")"); } while (0)
#define EXPECT_NEAR(a, b, tol) do { if (!(std::fabs(double(a) - double(b)) <= double(tol))) { std::printf("  %s:%d: EXPECT_NEAR(" #a ", " #b "): %g vs %g\n", __FILE__, __LINE__, double(a), double(b)); ++mini::failures(); } } while (0)
#define EXPECT_FLOAT_EQ(a, b) do { if (!mini::float_eq(float(a), float(b))) { std::printf("  %s:%d: EXPECT_FLOAT_EQ(" #a ", " #b "): %g vs %g\n", __FILE__, __LINE__, double(a), double(b)); ++mini::failures(); } } while (0)
#define EXPECT_GT(a, b) EXPECT_TRUE((a) > (b))
#define EXPECT_LT(a, b) EXPECT_TRUE((a) < (b))
#define EXPECT_GE(a, b) EXPECT_TRUE((a) >= (b))
#define EXPECT_LE(a, b) EXPECT_TRUE((a) <= (b))
#define EXPECT_THROW(stmt, ex) do { bool t_ = false; try { stmt; } catch (const ex&) { t_ = true; } catch (...) {} if (!t_) FAIL_MSG("EXPECT_THROW(" #stmt ")"); } while (0)
#define EXPECT_NO_THROW(stmt) do { try { stmt; } catch (...) { FAIL_MSG("EXPECT_NO_THROW(" #stmt ")"); } } while (0)
