"""GPU (-m gpu): pt_render_noalloc, the entry point for callers that cannot allocate device memory, through the
C-ABI against pt_render on the same inputs. Both run the same kernels with integer (order-independent) job sums,
so the bar is bit-exact floats and equal ray counts. Two calls in a row, the second with a smaller image: every
call on a GPU shares that GPU's fixed array, which keeps the previous call's contents.

Collected last on purpose (the round's end runs pytest -x)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(dev, torch, W, H, spp_begin, spp_end, depth, seed):
    ref = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rays_ref = dev.render(W, H, spp_begin, spp_end, depth, seed, accum_ptr=ref.data_ptr(), want_rays=True)
    out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    rays_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()
    dev.render_noalloc(W, H, spp_begin, spp_end, depth, seed, out.data_ptr(), rays_dev.data_ptr())
    dev.synchronize()
    return ref.cpu().numpy(), rays_ref, out.cpu().numpy(), int(rays_dev.item())


def test_noalloc_entry_point_matches_pt_render_twice_in_a_row():
    import torch
    from path_tracer_amd import Device, PtError, host
    dev = Device(0)
    try:
        dev.set_scene(host.generate_scene(250, 1, 2))
        dev.set_camera(host.camera_setup())
        # first call: a frame with partial tiles; second call: a smaller one (other plane offsets in the array)
        for (W, H, b, e, seed) in ((200, 113, 0, 6, 99), (96, 54, 3, 8, 7)):
            ref, rays_ref, out, rays = _both(dev, torch, W, H, b, e, 50, seed)
            assert rays == rays_ref and rays > W * H * (e - b)
            assert ref.any() and np.array_equal(out, ref), (W, H)
        # the call adds to the caller's buffer like pt_render, and works with the grid search too
        W, H = 96, 54
        buf = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        dev.render_noalloc(W, H, 0, 4, 50, 5, buf.data_ptr())
        dev.render_noalloc(W, H, 4, 8, 50, 5, buf.data_ptr())
        dev.synchronize()
        whole = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        dev.render(W, H, 0, 4, 50, 5, accum_ptr=whole.data_ptr())
        dev.render(W, H, 4, 8, 50, 5, accum_ptr=whole.data_ptr())
        dev.synchronize()
        assert np.array_equal(buf.cpu().numpy(), whole.cpu().numpy())
        dev.set_accel(1)
        ref, rays_ref, out, rays = _both(dev, torch, 120, 67, 0, 5, 50, 11)
        assert rays == rays_ref and np.array_equal(out, ref)
        dev.set_accel(0)
        # the largest image it serves, and one pixel more
        lib_max = 1920 * 1080
        big = torch.zeros((1080, 1920, 3), dtype=torch.float32, device="cuda")
        dev.render_noalloc(1920, 1080, 0, 1, 50, 1, big.data_ptr())
        dev.synchronize()
        ref = torch.zeros_like(big)
        dev.render(1920, 1080, 0, 1, 50, 1, accum_ptr=ref.data_ptr())
        dev.synchronize()
        assert np.array_equal(big.cpu().numpy(), ref.cpu().numpy())
        with pytest.raises(PtError, match="exceeds"):
            dev.render_noalloc(1921, 1080, 0, 1, 50, 1, big.data_ptr())
        assert 1921 * 1080 > lib_max
        with pytest.raises(PtError, match="caller passes"):
            dev.render_noalloc(64, 36, 0, 1, 50, 1, None)
        # and a small frame again after the large one
        ref, rays_ref, out, rays = _both(dev, torch, 64, 36, 0, 3, 50, 3)
        assert rays == rays_ref and np.array_equal(out, ref)
    finally:
        dev.close()


def test_noalloc_entry_point_allocates_nothing_and_can_be_captured():
    """The property the entry point exists for. (1) The device's free memory, as the driver reports it, does not move
    across calls (a hipMalloc of any size would show: the driver hands out whole pages). (2) The call can be recorded
    into a HIP graph (stream capture forbids allocation and synchronisation) and the graph replayed: each replay adds
    the same samples again, so two replays into a cleared buffer give exactly twice pt_render's sums."""
    import torch
    from path_tracer_amd import Device, host
    dev = Device(0)
    try:
        dev.set_scene(host.generate_scene(250, 1, 2))
        dev.set_camera(host.camera_setup())
        W, H, spp, seed = 160, 90, 4, 21
        out = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
        rays_dev = torch.zeros(1, dtype=torch.int64, device="cuda")
        ref = torch.zeros_like(out)
        rays_ref = dev.render(W, H, 0, spp, 50, seed, accum_ptr=ref.data_ptr(), want_rays=True)
        dev.render_noalloc(W, H, 0, spp, 50, seed, out.data_ptr(), rays_dev.data_ptr())          # warm: code objects loaded
        dev.synchronize(); torch.cuda.synchronize()
        free0, _ = torch.cuda.mem_get_info()
        for k in range(3):
            dev.render_noalloc(W, H, k, k + spp, 50, seed, out.data_ptr(), rays_dev.data_ptr())
        dev.synchronize()
        free1, _ = torch.cuda.mem_get_info()
        assert free1 == free0, (free0, free1)

        side = torch.cuda.Stream()
        dev.set_stream(side.cuda_stream)                     # before the capture begins: pt_set_stream synchronises
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
            dev.render_noalloc(W, H, 0, spp, 50, seed, out.data_ptr(), rays_dev.data_ptr())
        torch.cuda.synchronize()
        out.zero_(); rays_dev.zero_()
        torch.cuda.synchronize()
        graph.replay(); graph.replay()
        torch.cuda.synchronize()
        assert int(rays_dev.item()) == rays_ref                                    # each replay counts its own rays
        assert ref.any() and torch.equal(out, 2 * ref)
        del graph
        dev.set_stream(None)
    finally:
        dev.close()
