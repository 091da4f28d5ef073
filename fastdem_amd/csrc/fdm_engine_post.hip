// fdm_engine_post.hip — the stages behind the hot path: stencil post-processing (fdm_post.hpp),
// map egress (fdm_egress.hpp), PointCloud2 ingest (fdm_ingest.hpp).  One of the library's three translation units
// (fdm_engine_host.hpp).
#include "fdm_engine_host.hpp"

#include "fdm_engine_post.inl"
#include "fdm_engine_io.inl"
