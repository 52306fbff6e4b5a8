"""fsa_sort_particle_static (the counting sort on caller-owned arrays, no allocation) against the solver's own
fsa_calc_hash + fsa_sort_particle: hash and permutation bit for bit, on two calls in a row, the second one
smaller (the library's static work array keeps the first call's contents)."""
import numpy as np
import pytest

from fluidsolver_amd import capi, scenes

pytestmark = pytest.mark.gpu


def _solver_sort(scene):
    p = capi.make_params(capi.PRECISION_F32, **scene["params"])
    with capi.Solver(p, len(scene["pos"])) as s:
        s.upload(scene)
        s.calc_hash()
        s.sort_particle()
        return p, s.get("particleHash"), s.get("particleIndex")


def _static_sort(p, pos):
    import torch
    n = len(pos)
    d_pos = torch.from_numpy(np.ascontiguousarray(pos, np.float32)).cuda()
    d_hash = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    d_index = torch.full((max(n, 1),), -7, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    capi.sort_particle_static(p, n, d_pos.data_ptr(), d_hash.data_ptr(), d_index.data_ptr(), device=0,
                              stream_ptr=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return d_hash.cpu().numpy()[:n], d_index.cpu().numpy()[:n]


def _scene(nx, ny, nz, seed, outside):
    sc = scenes.jittered(scenes.dam_break(nx, ny, nz, tank_x_factor=2.0), seed=seed)
    # some particles on the faces of the grid box and outside it: GRID_UNDEF, stable tail
    rng = np.random.default_rng(seed)
    pos = sc["pos"].copy()
    vmin, vmax = np.array(sc["params"]["vmin"], np.float32), np.array(sc["params"]["vmax"], np.float32)
    idx = rng.choice(len(pos), size=outside, replace=False)
    pos[idx[: outside // 2]] = vmax + np.float32(0.01)
    pos[idx[outside // 2:], 0] = vmin[0]
    sc["pos"] = pos
    return sc


def test_static_sort_matches_solver_twice():
    big = _scene(40, 24, 32, seed=5, outside=37)
    small = _scene(12, 10, 8, seed=6, outside=5)
    for sc in (big, small):  # the second call is the smaller one
        p, h_ref, i_ref = _solver_sort(sc)
        assert (h_ref == capi.GRID_UNDEF).sum() >= 5
        h, i = _static_sort(p, sc["pos"])
        assert np.array_equal(h, h_ref)
        assert np.array_equal(i, i_ref)


def test_static_sort_allocates_nothing_and_runs_under_stream_capture():
    """what the entry point exists for: after the first call on a GPU a call changes no free device memory, and it
    can be recorded into a graph under a capture that rejects every allocation (global capture mode), whose replay
    gives the same hash and permutation"""
    import torch
    sc = _scene(24, 16, 12, seed=9, outside=11)
    p, h_ref, i_ref = _solver_sort(sc)
    n = len(sc["pos"])
    d_pos = torch.from_numpy(np.ascontiguousarray(sc["pos"], np.float32)).cuda()
    d_hash = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    d_index = torch.full((n,), -7, dtype=torch.int32, device="cuda")
    side = torch.cuda.Stream()

    def call(stream):
        capi.sort_particle_static(p, n, d_pos.data_ptr(), d_hash.data_ptr(), d_index.data_ptr(), device=0,
                                  stream_ptr=stream.cuda_stream)

    call(side)                       # first call on this GPU (it may be the process's first): address look-up
    side.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    call(side)
    side.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free1 == free0
    assert np.array_equal(d_hash.cpu().numpy(), h_ref) and np.array_equal(d_index.cpu().numpy(), i_ref)

    d_hash.fill_(-7)
    d_index.fill_(-7)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side, capture_error_mode="global"):
        call(torch.cuda.current_stream())
    torch.cuda.synchronize()
    assert int(d_hash[0]) == -7      # captured, not run
    g.replay()
    torch.cuda.synchronize()
    assert np.array_equal(d_hash.cpu().numpy(), h_ref) and np.array_equal(d_index.cpu().numpy(), i_ref)
    assert torch.cuda.current_device() == 0


def test_static_sort_limits_and_empty():
    import torch
    sc = scenes.dam_break(8, 8, 8)
    p = capi.make_params(capi.PRECISION_F32, **sc["params"])
    buf = torch.zeros(16, dtype=torch.int32, device="cuda")
    lib = capi.load()
    with pytest.raises(capi.FsaError, match="at most"):
        capi.sort_particle_static(p, capi.SORT_STATIC_MAX_PARTICLES + 1, buf.data_ptr(), buf.data_ptr(), buf.data_ptr())
    big = capi.make_params(capi.PRECISION_F32, spacing=0.005, smoothratio=2.0, vmin=(-1.0, -1.0, -1.0),
                           vmax=(1.0, 1.0, 1.0))  # 200^3 cells
    assert big.num_cells > capi.SORT_STATIC_MAX_CELLS
    with pytest.raises(capi.FsaError, match="cells"):
        capi.sort_particle_static(big, 4, buf.data_ptr(), buf.data_ptr(), buf.data_ptr())
    with pytest.raises(capi.FsaError, match="NULL"):
        capi.sort_particle_static(p, 4, 0, buf.data_ptr(), buf.data_ptr())
    capi.sort_particle_static(p, 0, 0, 0, 0)  # nothing to do
    assert lib is not None
