// fdm_engine_tiled.hip — host side of the LARGE-scan pipeline (fdm_tbin2.hpp + fdm_tiled.hpp): the record pools, which tiles
// the update wavefronts walk, k_tbin / k_tupdate on their own and the fused k_tupdate_tbin.  One of the library's
// translation units (fdm_engine_host.hpp): an A/B of these kernels rebuilds this unit only.
#include "fdm_engine_host.hpp"

namespace fdmh {

// Which tiles the update wavefronts of a launch walk (TileWork, fdm_tiled.hpp): `blocks` update blocks of four
// wavefronts each, tiles dealt round robin.
TileWork tile_work(const fdm_engine* e, unsigned blocks) {
  TileWork K{};
  K.W = blocks * 4u;
  K.T = (e->TG.n_tiles + K.W - 1u) / K.W;
  K.prio = e->upd_prio ? 1u : 0u;
  K.stagger = unsigned(e->bin_stagger);
  K.delay = unsigned(e->bin_delay);
  K.delay_blocks = unsigned(e->bin_delay_blocks);
  return K;
}
// update blocks of a launch: alone, enough to fill the chip; beside a bin half (fused launch), few — the bin blocks are
// the arithmetic, the update wavefronts are chains of round trips that run beside them (option "upd_blocks")
unsigned update_blocks(const fdm_engine* e, bool fused) {
  const unsigned most = (e->TG.n_tiles + 3u) / 4u;  // one tile per wavefront
  const unsigned want = fused ? unsigned(e->upd_blocks) : unsigned(e->upd_blocks_alone);
  return std::max(1u, std::min(most, want));
}

// What the update wavefronts keep between scans (stamps, statistics, rare-path scratch).
int ensure_tile_aux(fdm_engine* e) {
  if (!e->tile_stamp32) {
    e->TG.tiles_r = (e->G.s_rows + kTS - 1) / kTS;
    e->TG.tiles_c = (e->G.s_cols + kTC - 1) / kTC;
    e->TG.n_tiles = unsigned(e->TG.tiles_r) * unsigned(e->TG.tiles_c);
    if (e->TG.n_tiles >= (1u << 23)) return fail(FDM_ERR_INVALID, "tiled pipeline: more than 2^23 map tiles");
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->tile_stamp32), e->TG.n_tiles * sizeof(uint32_t)));
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->upd_part32), e->TG.n_tiles * sizeof(uint32_t)));
    hipLaunchKernelGGL(k_fill_u32, dim3(64), dim3(256), 0, e->stream, e->tile_stamp32, 0xFFFFFFFEu, size_t(e->TG.n_tiles));
    hipLaunchKernelGGL(k_fill_u32, dim3(64), dim3(256), 0, e->stream, e->upd_part32, 0u, size_t(e->TG.n_tiles));
    HIPCK(hipGetLastError());
  }
  const size_t waves = 4u * size_t(std::max(update_blocks(e, false), update_blocks(e, true)));
  if (!e->tile_rare || waves > e->tile_rare_waves) {  // the update's rare-path scratch: 3 KB per update wavefront
    if (e->tile_rare) {
      if (int rc_sync = sync_all(e)) return rc_sync;
      HIPCK(hipFree(e->tile_rare));
      e->tile_rare = nullptr;
    }
    HIPCK(hipMalloc(reinterpret_cast<void**>(&e->tile_rare), waves * 3u * kTileCells * sizeof(uint32_t)));
    e->tile_rare_waves = waves;
    for (auto& q : e->pool) q.rare = e->tile_rare;
  }
  return FDM_OK;
}

// The record pools of the tiled pipeline: `records` per pool, `blocks` chunk slots per tile.
int ensure_tile_pool(fdm_engine* e, size_t records, unsigned blocks, bool has_int, bool has_col) {
  if (int rc_aux = ensure_tile_aux(e)) return rc_aux;
  const bool grow_rec = records > e->pool_cap;
  (void)has_int; (void)has_col;
  const bool grow_desc = blocks > e->desc_stride;
  if (!grow_rec && !grow_desc) return FDM_OK;
  if (int rc_sync = sync_all(e)) return rc_sync;  // (every chunk list is consumed: the tile counters are all zero)
  if (grow_rec) {
    e->pool_cap = records + records / 4 + 8192;
    if (e->pool_cap >= 0x7FFFFFF0ull) return fail(FDM_ERR_INVALID, "tiled pipeline: scan too large");
  }
  if (grow_desc) e->desc_stride = blocks + blocks / 4 + 16;
  for (auto& q : e->pool) {
    auto re = [&](auto*& ptr, size_t bytes) -> int {
      if (ptr) HIPCK(hipFree(ptr));
      ptr = nullptr;
      HIPCK(hipMalloc(reinterpret_cast<void**>(&ptr), bytes));
      return FDM_OK;
    };
    int rc;
    if (grow_rec) {
      if ((rc = re(q.hot, e->pool_cap * sizeof(RecHot)))) return rc;
      if ((rc = re(q.cold, e->pool_cap * sizeof(RecCold)))) return rc;
    }
    if (grow_desc && (rc = re(q.desc, size_t(e->TG.n_tiles) * e->desc_stride * 8))) return rc;  // (only entries below a tile's counter are ever read)
    if (!q.cnt) {
      q.cnt_shift = unsigned(e->cnt_shift);
      HIPCK(hipMalloc(reinterpret_cast<void**>(&q.cnt), (size_t(e->TG.n_tiles) << q.cnt_shift) * sizeof(unsigned)));
      HIPCK(hipMemsetAsync(q.cnt, 0, (size_t(e->TG.n_tiles) << q.cnt_shift) * sizeof(unsigned), e->stream));
    }
    q.stride = e->desc_stride;
  }
  return FDM_OK;
}

// dynamic LDS of a large-scan launch that needs `lds` bytes: padded so that six blocks share a CU's 160 KB, not seven
unsigned tiled_lds_padded(const fdm_engine* e, unsigned lds) {
  if (e->tiled_lds_pad >= 0) return lds + unsigned(e->tiled_lds_pad);
  constexpr unsigned kSeven = 163840u / 7u;  // at most this much: seven blocks fit
  return lds <= kSeven ? kSeven + 16u : lds;
}
int launch_tbin(fdm_engine* e, const ScanParams& P, const ScanInputs& in, const TilePool& Q, int32_t* ids,
                unsigned bin_blocks, fdm_engine::BinVariant bv) {
  const unsigned lds = tiled_lds_padded(e, tbin_lds_bytes(bv.has_int, bv.has_col, bv.threads));
  int rc = FDM_OK;
  auto go = [&](auto kern) {
    if ((rc = allow_lds(kern, lds))) return;
    hipLaunchKernelGGL(kern, dim3(bin_blocks), dim3(bv.threads), lds, e->stream, P, e->G, e->TG, e->d_state, in, e->S,
                       Q, ids);
  };
#define FDM_TBIN(LN)                                               \
  if (bv.has_int && bv.has_col) go(k_tbin<true, true, 256, LN>);    \
  else if (bv.has_int) go(k_tbin<true, false, 256, LN>);            \
  else if (bv.has_col) go(k_tbin<false, true, 256, LN>);            \
  else go(k_tbin<false, false, 256, LN>);
  if (bv.lean == 1) { FDM_TBIN(true) } else { FDM_TBIN(false) }
#undef FDM_TBIN
  if (rc) return rc;
  HIPCK(hipGetLastError());
  return FDM_OK;
}


// The held-back (or just enqueued) update of a large scan on its own.
int launch_tiled_update_alone(fdm_engine* e, const fdm_engine::PendingUpdate& u) {
  return with_policy(e, [&](auto tag, const auto& layers) -> int {
    using POLICY = decltype(tag);
    if constexpr (is_rec_policy<POLICY>) {
      const unsigned blocks = update_blocks(e, false);
      const TileWork K = tile_work(e, blocks);
      const bool hi = u.P.has_intensity != 0, hc = u.P.has_color != 0;
      const unsigned lds = tile_lds_bytes(hi, hc);
      auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, e->stream, u.P, e->G, e->TG, e->d_state, layers,
                           e->d_layer_ptrs, e->n_layer_ptrs, u.Q, u.A, K);
      };
      if (hi && hc) go(k_tupdate<POLICY, true, true>);
      else if (hi) go(k_tupdate<POLICY, true, false>);
      else if (hc) go(k_tupdate<POLICY, false, true>);
      else go(k_tupdate<POLICY, false, false>);
    } else {
      return fail(FDM_ERR_INVALID, "internal: tiled update with a per-layer policy");
    }
    HIPCK(hipGetLastError());
    return FDM_OK;
  });
}

// The held-back update of large scan t and the bin of scan t + 1 in one launch.
int launch_tiled_update_fused(fdm_engine* e, const fdm_engine::PendingUpdate& u, const ScanParams& Pb,
                              const ScanInputs& Ib, const TilePool& Qb, int32_t* ids_b, unsigned bin_blocks_b,
                              fdm_engine::BinVariant bv) {
  const Scratch Sb = e->S;
  return with_policy(e, [&](auto tag, const auto& layers) -> int {
    using POLICY = decltype(tag);
    constexpr bool kRec = is_rec_policy<POLICY>;
    {
      if constexpr (kRec) {
        const unsigned ub = update_blocks(e, true);
        const TileWork K = tile_work(e, ub);
        const unsigned lds = tiled_lds_padded(e, std::max(tile_lds_bytes(bv.has_int, bv.has_col), tbin_lds_bytes(bv.has_int, bv.has_col, bv.threads)));
        int rc = FDM_OK;
        auto go = [&](auto kern) {
          if ((rc = allow_lds(kern, lds))) return;
          TileAux A = u.A;
          if (ub + bin_blocks_b > e->timeline_cap) A.timeline = nullptr;
          e->timeline_blocks = A.timeline ? ub + bin_blocks_b : 0u;
          e->timeline_upd = ub;
          hipLaunchKernelGGL(kern, dim3(ub + bin_blocks_b), dim3(bv.threads), lds, e->stream, u.P, e->G, e->TG,
                             e->d_state, layers, e->d_layer_ptrs, e->n_layer_ptrs, u.Q, A, K, ub, Pb, Ib, Sb, Qb,
                             ids_b);
        };
#define FDM_TF(LN)                                                                  \
        if (bv.has_int && bv.has_col) go(k_tupdate_tbin<POLICY, true, true, 256, LN>);   \
        else if (bv.has_int) go(k_tupdate_tbin<POLICY, true, false, 256, LN>);           \
        else if (bv.has_col) go(k_tupdate_tbin<POLICY, false, true, 256, LN>);           \
        else go(k_tupdate_tbin<POLICY, false, false, 256, LN>);
        if (bv.lean == 1) { FDM_TF(true) } else { FDM_TF(false) }
#undef FDM_TF
        if (rc) return rc;
      } else {
        return fail(FDM_ERR_INVALID, "internal: tiled update with a per-layer policy");
      }
      HIPCK(hipGetLastError());
      return FDM_OK;
    }
    return FDM_OK;
  });
}

}  // namespace fdmh
