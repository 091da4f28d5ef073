// fdm_engine_launch.inl — launches of the SMALL-scan path (per-cell scratch kernels k_update* / k_update_bin*) on their own and
// fused with the next scan's bin half; the large-scan launches live in fdm_engine_tiled.hip, the batch launches in
// fdm_engine_multi.hip.  Part of fdm_engine.hip (inside its namespace fdmh): do not compile on its own.

// The held-back (or just enqueued) update on its own.
int launch_update_alone(fdm_engine* e, const fdm_engine::PendingUpdate& u) {
  if (u.multi) return launch_multi_update(e, u);
  if (u.tiled) return launch_tiled_update_alone(e, u);  // (fdm_engine_tiled.hip)
  return with_policy(e, [&](auto tag, const auto& layers) -> int {
    using POLICY = decltype(tag);
    if (u.S.dense) {
      hipLaunchKernelGGL(k_update<POLICY>, dim3(u.upd_blocks), dim3(256), 0, e->stream, u.P, e->G, e->d_state, layers,
                         e->d_layer_ptrs, e->n_layer_ptrs, u.S, u.in.x, u.in.y, u.in.z, u.in.intensity, u.in.rgb,
                         u.in.var, unsigned(e->ncell));
    } else {  // stamp-gated: kStampTiles tiles per block, idle tiles cost one scalar load
      hipLaunchKernelGGL(k_update_stamped<POLICY>, dim3((u.upd_blocks + kStampTiles - 1) / kStampTiles), dim3(256), 0,
                         e->stream, u.P, e->G, e->d_state, layers, e->d_layer_ptrs, e->n_layer_ptrs, u.S, u.in.x,
                         u.in.y, u.in.z, u.in.intensity, u.in.rgb, u.in.var, unsigned(e->ncell));
    }
    HIPCK(hipGetLastError());
    return FDM_OK;
  });
}

// The held-back update of scan t and the bin of scan t+1 in one launch.
int launch_update_fused(fdm_engine* e, const fdm_engine::PendingUpdate& u, const ScanParams& Pb,
                        const ScanInputs& Ib, const TilePool& Qb, int32_t* ids_b, unsigned bin_blocks_b,
                        fdm_engine::BinVariant bv) {
  if (u.tiled) return launch_tiled_update_fused(e, u, Pb, Ib, Qb, ids_b, bin_blocks_b, bv);  // (fdm_engine_tiled.hip)
  const Scratch Sb = e->S;
  return with_policy(e, [&](auto tag, const auto& layers) -> int {
    using POLICY = decltype(tag);
    constexpr bool kRec = is_rec_policy<POLICY>;
    auto go = [&](auto kern, unsigned threads) {
      // tiles per update block: threads / 256, times kStampTiles slots on stamp-gated maps
      const unsigned per = (threads / 256u) * (u.S.dense ? 1u : kStampTiles);
      const unsigned ub = (u.upd_blocks + per - 1u) / per;
      hipLaunchKernelGGL(kern, dim3(ub + bin_blocks_b), dim3(threads), 0, e->stream, u.P, e->G, e->d_state, layers,
                         e->d_layer_ptrs, e->n_layer_ptrs, u.S, u.in, unsigned(e->ncell), ub, Pb, Sb, Ib, ids_b);
    };
    if (!bv.bin4) {
      if (!u.S.dense) {  // stamp-gated maps: one generic variant
        go(k_update_bin<POLICY, true, true>, 256u);
      } else if (kRec && bv.lean == 1) {  // channel tests folded at compile time, optional work compiled out
        if constexpr (kRec) {
          if (bv.has_int && bv.has_col) go(k_update_bin<POLICY, true, false, 3, 1>, 256u);
          else if (bv.has_col) go(k_update_bin<POLICY, true, false, 2, 1>, 256u);
          else if (bv.has_int) go(k_update_bin<POLICY, true, false, 1, 1>, 256u);
          else go(k_update_bin<POLICY, true, false, 0, 1>, 256u);
        }
      } else if (kRec && bv.lean == 2) {  // ... but x / y / z written through to the engine's staging block
        if constexpr (kRec) {
          if (bv.has_int && bv.has_col) go(k_update_bin<POLICY, true, false, 3, 2>, 256u);
          else if (bv.has_col) go(k_update_bin<POLICY, true, false, 2, 2>, 256u);
          else if (bv.has_int) go(k_update_bin<POLICY, true, false, 1, 2>, 256u);
          else go(k_update_bin<POLICY, true, false, 0, 2>, 256u);
        }
      } else {  // everything else (captures, cell ids, per-layer layout ...): channels read from ScanParams
        go(k_update_bin<POLICY, true, false>, 256u);
      }
    } else if constexpr (kRec) {
      if (!u.S.dense) return fail(FDM_ERR_INVALID, "internal: k_bin4 fused with a stamp-gated update");
#define FDM_FUSED4(LN)                                                                       \
      if (bv.has_int && bv.has_col) go(k_update_bin4<POLICY, true, true, 256, false, LN>, 256u);   \
      else if (bv.has_int) go(k_update_bin4<POLICY, true, false, 256, false, LN>, 256u);           \
      else if (bv.has_col) go(k_update_bin4<POLICY, false, true, 256, false, LN>, 256u);           \
      else go(k_update_bin4<POLICY, false, false, 256, false, LN>, 256u);
      if (bv.lean == 1) { FDM_FUSED4(1) } else if (bv.lean == 2) { FDM_FUSED4(2) } else { FDM_FUSED4(0) }
#undef FDM_FUSED4
    } else {
      return fail(FDM_ERR_INVALID, "internal: k_bin4 fused with a per-layer policy");
    }
    HIPCK(hipGetLastError());
    return FDM_OK;
  });
}
