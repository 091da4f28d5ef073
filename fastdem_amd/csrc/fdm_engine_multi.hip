// fdm_engine_multi.hip — host side of the batch pipeline (fdm_multi.hpp k_mbatch, fdm_rbatch.hpp): which scans of a call leave
// as one batch, the per-batch scratch sets, the launches.  One of the library's translation units (fdm_engine_host.hpp).
#include "fdm_engine_host.hpp"

namespace fdmh {
#include "fdm_engine_multi.inl"
}  // namespace fdmh
