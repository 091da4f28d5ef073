"""fdm_host_alloc / fdm_host_free (include/fdm_engine.h): pooled pinned memory for input clouds.
Without a GPU the blocks are ordinary memory (the engine would copy them); on a GPU box they are
pinned, pooled, and read in place by the synchronous host entry points too."""
import numpy as np
import pytest

F32 = np.float32


def test_blocks_without_a_gpu_are_plain_memory():
    import torch
    import fastdem_amd
    if torch.cuda.is_available():
        pytest.skip("pageable fallback is only taken without a GPU")
    h = fastdem_amd.host_array(np.arange(1000, dtype=F32))
    assert not h.pinned
    assert np.array_equal(h.array, np.arange(1000, dtype=F32))
    lib = fastdem_amd.capi.load()
    lib.fdm_host_free(None)
    lib.fdm_host_trim()
    del h


def test_zero_length_block():
    import fastdem_amd
    h = fastdem_amd.HostArray(0)
    assert h.array.size == 0


@pytest.mark.gpu
def test_pool_reuses_pinned_blocks(gpu):
    lib = gpu.capi.load()
    a = lib.fdm_host_alloc(300000)
    assert lib.fdm_host_is_pinned(a) == 1
    lib.fdm_host_free(a)
    b = lib.fdm_host_alloc(270000)  # same 512 KiB class
    assert a == b
    lib.fdm_host_free(b)
    lib.fdm_host_trim()
    big = lib.fdm_host_alloc((1 << 30) + 4096)  # above the largest class: not pooled
    assert big and lib.fdm_host_is_pinned(big) == 1
    lib.fdm_host_free(big)


@pytest.mark.gpu
@pytest.mark.parametrize("entry", ["integrate", "update"])
def test_synchronous_entry_points_read_pinned_arrays_in_place(gpu, R, entry):
    """fdm_engine_integrate / fdm_engine_update on pooled pinned arrays (no copy commands) against the
    oracle, stats included; odd sizes and a z-variance channel exercise the write-through of the
    channels only the update kernel consumes."""
    from helpers import assert_layers_equal, pair, same_geometry
    wl = gpu.synth.vlp16(n_scans=3)
    eng, ref = pair(gpu, R, wl.width, wl.height, wl.resolution, wl.apply_to)
    rng = np.random.default_rng(3)
    for k in range(3):
        s = wl.scan(k)
        n = s["x"].size - 7 * k - 1
        ch = {c: gpu.host_array(s[c][:n]) for c in ("x", "y", "z", "intensity")}
        assert all(h.pinned for h in ch.values())
        if entry == "integrate":
            out_e = eng.integrate(ch["x"].array, ch["y"].array, ch["z"].array, wl.T_base_sensor, wl.pose(k),
                                  intensity=ch["intensity"].array)
            out_r = ref.integrate(s["x"][:n], s["y"][:n], s["z"][:n], wl.T_base_sensor, wl.pose(k),
                                  intensity=s["intensity"][:n])
        else:
            var = gpu.host_array(rng.uniform(1e-4, 5e-3, n).astype(F32))
            out_e = eng.update(ch["x"].array, ch["y"].array, ch["z"].array, (0.1 * k, -0.2 * k), z_var=var.array,
                               intensity=ch["intensity"].array)
            out_r = ref.update(s["x"][:n], s["y"][:n], s["z"][:n], (0.1 * k, -0.2 * k), z_var=var.array.copy(),
                               intensity=s["intensity"][:n])
        assert out_e == out_r
    assert_layers_equal(eng, ref)
    assert same_geometry(eng.geometry(), ref.geometry())
