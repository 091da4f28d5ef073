// fdm_engine_ray.hip — the raycasting stage of one scan: voxel filter, stable radix sort, ray queue,
// the walks (fdm_raycast.hpp, fdm_raywedge.hpp, fdm_rsort.hpp), resolve; entry points fdm_engine_apply_raycasting*,
// fdm_engine_voxel_any, fdm_engine_last_ray_ms.  One of the library's three translation units (fdm_engine_host.hpp).
#include "fdm_engine_host.hpp"

#include "fdm_engine_ray.inl"
