"""Initializer network: the torch module against the fp64 NumPy forward of the
same weights, the input/output glue, and the reference-shaped NNPlanner / NeoPlanner plumbing.
CPU here; `test_initializer_on_gpu_feeds_the_optimiser` runs the same on the MI355X."""
import numpy as np
import pytest
import torch

from neo_planner_amd import initializer as ini
from oracle import plannernet_np as pnp


def _net(h=48, w=64, seed=0):
    torch.manual_seed(seed)
    net = ini.PlannerNet(img_height=h, img_width=w).eval()
    # non-trivial BatchNorm statistics, as a trained net would have
    for m in net.modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            m.running_mean.normal_(0, 0.2)
            m.running_var.uniform_(0.5, 1.5)
            m.weight.data.uniform_(0.5, 1.5)
            m.bias.data.normal_(0, 0.1)
    return net


def test_state_dict_uses_the_reference_parameter_names():
    keys = set(ini.PlannerNet().state_dict().keys())
    for k in ("img_backbone.conv1.weight", "img_backbone.bn1.running_mean", "img_backbone.layer2.0.downsample.0.weight",
              "img_backbone.layer4.1.conv2.weight", "img_backbone.fc.bias", "motion_backbone.0.weight",
              "motion_backbone.6.bias", "mlp.0.weight", "mlp.6.weight"):
        assert k in keys
    sd = ini.PlannerNet().state_dict()
    assert tuple(sd["img_backbone.conv1.weight"].shape) == (64, 1, 7, 7)
    assert tuple(sd["img_backbone.fc.weight"].shape) == (24, 512)
    assert tuple(sd["mlp.0.weight"].shape) == (48, 48) and tuple(sd["mlp.6.weight"].shape) == (9, 96)
    n_dense = sum(sd[f"{blk}.{i}.weight"].numel() for blk, idx in (("motion_backbone", (0, 2, 4, 6)), ("mlp", (0, 2, 4, 6))) for i in idx)
    assert n_dense == 20448 + 0          # MACs per trajectory of the dense part


def test_torch_forward_matches_numpy_fp64_forward():
    h, w = 48, 64
    net = _net(h, w).double()
    p = {k: v.numpy().astype(np.float64) for k, v in net.state_dict().items()}
    rng = np.random.default_rng(0)
    inp = np.concatenate([rng.integers(0, 256, (3, h * w)).astype(np.float64), rng.normal(0, 1, (3, 24))], axis=1)
    with torch.no_grad():
        out = net(torch.from_numpy(inp)).numpy()
    ref = pnp.forward(inp, p, h, w)
    assert out.shape == (3, 9)
    assert np.max(np.abs(out - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))
    # split path (backbone once per scene, dense head per trajectory) == monolithic forward
    with torch.no_grad():
        f = net.image_features(torch.from_numpy(inp[:1, :h * w].reshape(1, 1, h, w)))
        out2 = net.head(f, torch.from_numpy(inp[:, h * w:])).numpy()
    ref2 = pnp.head(pnp.image_features(inp[:1, :h * w].reshape(1, 1, h, w), p), inp[:, h * w:], p)
    assert np.max(np.abs(out2 - ref2)) <= 1e-9 * max(1.0, np.max(np.abs(ref2)))


def test_quaternion_helper():
    q = ini.Quat.from_yaw(0.7)
    R = q.rotation_matrix
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15) and np.isclose(np.linalg.det(R), 1.0)
    v = np.array([1.0, -2.0, 0.5])
    assert np.allclose(q.inverse.rotate(q.rotate(v)), v, atol=1e-15)
    assert np.allclose(q.rotate([1, 0, 0]), [np.cos(0.7), np.sin(0.7), 0.0])


def test_form_nn_input_layout():
    ds = ini.DroneState()
    ds.global_pos = np.array([1.0, 2.0, 2.0]); ds.global_vel = np.array([0.5, 0.0, 0.0])
    ds.attitude = ini.Quat.from_yaw(np.pi / 2); ds.local_vel = np.array([0.0, -0.5, 0.0])
    start = ini.DroneState(); start.global_pos = np.array([1.5, 2.0, 2.0]); start.global_vel = np.array([0.6, 0.1, 0.0])
    target = np.array([[1.0, 7.0], [0.0, 0.8]])
    depth = np.linspace(0.5, 9.0, 12).reshape(3, 4)
    img, motion = ini.form_nn_input(depth, ds, 2.0, start, target)
    assert img.dtype == np.uint8 and img.max() == 255 and img.shape == (3, 4)
    assert motion.shape == (24,)
    assert np.allclose(motion[:3], ds.local_vel)
    assert np.allclose(motion[3:12], ds.attitude.rotation_matrix.reshape(-1))
    # target 5 m ahead along world +y = body +x for a 90 degree yaw
    assert np.allclose(motion[18:21], [5.0, 0.0, 0.0], atol=1e-12)
    flat = ini.process_input_np(img, motion)
    assert flat.dtype == np.float32 and flat.shape == (12 + 24,)


def test_nn_planner_output_goes_to_world_frame():
    h, w = 48, 64
    nn_pl = ini.NNPlanner(des_pos_z=2.0, net=_net(h, w), device="cpu")
    ds = ini.DroneState(); ds.global_pos = np.array([2.0, -1.0, 2.0]); ds.attitude = ini.Quat.from_yaw(0.3)
    target = np.array([[7.0, -1.0], [0.8, 0.0]])
    rng = np.random.default_rng(1)
    depth = rng.uniform(0.3, 8.0, (h, w))
    nn_pl.nn_traj_plan(depth, ds, ds, target)
    assert nn_pl.int_wpts.shape == (2, 2) and nn_pl.ts.shape == (3,)
    img, motion = ini.form_nn_input(depth, ds, 2.0, ds, target)
    with torch.no_grad():
        out = nn_pl.net(torch.from_numpy(ini.process_input_np(img, motion))[None])[0].numpy()
    local, ts = ini.split_output(out)
    want = np.stack([ds.attitude.rotate(local[:, i]) + ds.global_pos for i in range(2)], axis=1)[:2]
    assert np.allclose(nn_pl.int_wpts, want, atol=1e-6) and np.allclose(nn_pl.ts, ts)


def test_batch_initializer_matches_single_path():
    h, w = 48, 64
    net = _net(h, w)
    bi = ini.BatchInitializer(net=net, device="cpu")
    rng = np.random.default_rng(2)
    depth = rng.integers(0, 256, (h, w)).astype(np.uint8)
    B = 5
    motion = rng.normal(0, 1, (B, 24))
    yaws = rng.uniform(-1, 1, B)
    R = np.stack([ini.Quat.from_yaw(y).rotation_matrix for y in yaws])
    pos = rng.uniform(-2, 2, (B, 3))
    wp, ts = bi.warm_start(bi.scene_feature(depth), motion, R, pos, clamp_ts=False)
    assert wp.shape == (B, 2, 2) and ts.shape == (B, 3)
    for b in range(B):
        with torch.no_grad():
            out = net(torch.from_numpy(ini.process_input_np(depth, motion[b]))[None])[0].numpy()
        local, t1 = ini.split_output(out)
        want = np.stack([R[b] @ local[:, i] + pos[b] for i in range(2)], axis=1)[:2]
        assert np.allclose(wp[b].numpy(), want, atol=1e-5) and np.allclose(ts[b].numpy(), t1, atol=1e-6)
    _, ts_c = bi.warm_start(bi.scene_feature(depth), motion, R, pos)
    assert float(ts_c.min()) > 0.5 and float(ts_c.max()) < 5.0


@pytest.mark.gpu
def test_initializer_on_gpu_feeds_the_optimiser():
    """cfg3 data flow on the MI355X: one backbone pass per scene, dense head for B trajectories, the
    warm starts go straight into the batched optimiser; fp32 GPU forward vs the fp64 NumPy forward."""
    import neo_planner_amd as npa
    from neo_planner_amd import synth
    h, w = 120, 160
    net = _net(h, w)
    bi = ini.BatchInitializer(net=net, device="cuda")
    rng = np.random.default_rng(3)
    depth = rng.integers(0, 256, (h, w)).astype(np.uint8)
    B = 256
    hd, tl, _, _ = synth.replan_requests(4, B, 2, D=2, length_range=(4.0, 6.0))
    motion = rng.normal(0, 0.3, (B, 24))
    R = np.tile(np.eye(3), (B, 1, 1))
    pos = np.concatenate([hd[:, 0], np.full((B, 1), 2.0)], axis=1)
    feat = bi.scene_feature(depth)
    p = {k: v.cpu().numpy().astype(np.float64) for k, v in net.state_dict().items()}
    ref = pnp.head(pnp.image_features(depth.astype(np.float64).reshape(1, 1, h, w), p), motion, p)
    with torch.no_grad():
        out = net.head(feat, torch.as_tensor(motion, dtype=torch.float32, device="cuda")).cpu().numpy()
    assert np.max(np.abs(out - ref)) <= 2e-3 * max(1.0, np.max(np.abs(ref)))     # fp32 convolutions (MIOpen)
    wp, ts = bi.warm_start(feat, motion, R, pos)
    # spread the (untrained) network's waypoints between start and goal so the optimiser has work to do
    wp = wp.cpu().numpy() * 0.05 + np.stack([hd[:, 0] + (tl[:, 0] - hd[:, 0]) * f for f in (1 / 3, 2 / 3)], axis=2)
    occ = synth.occupancy_2d(4)
    m = npa.ESDF(); m.occupancy_map_cb(synth.OccupancyGridMsg(occ))
    bp = npa.BatchPlanner()
    res = bp.optimize(m, bp.pack_x(wp, ts.cpu().numpy()), hd, tl)
    # (an untrained network hands over poor durations: some runs end where the reference raises OverflowError)
    assert (res["status"] <= 2).mean() > 0.75 and np.all(np.isfinite(res["final_cost"][res["status"] <= 2]))


def test_conv1d_variant_matches_numpy_fp64_forward_and_reference_names():
    """nn_trainer_conv.py:107-159: Conv1d motion branch and fusion head; same Sequential indices as the reference"""
    h, w = 48, 64
    torch.manual_seed(3)
    net = ini.PlannerNetConv(img_height=h, img_width=w).eval().double()
    sd = net.state_dict()
    for k, shape in (("motion_backbone.0.weight", (16, 1, 3)), ("motion_backbone.2.weight", (32, 16, 3)),
                     ("motion_backbone.4.weight", (64, 32, 3)), ("motion_backbone.7.weight", (24, 64 * 24)),
                     ("mlp.0.weight", (16, 1, 3)), ("mlp.4.bias", (64,)), ("mlp.7.weight", (9, 64 * 48)),
                     ("img_backbone.fc.weight", (24, 512))):
        assert tuple(sd[k].shape) == shape, k
    p = {k: v.numpy().astype(np.float64) for k, v in sd.items()}
    rng = np.random.default_rng(1)
    inp = np.concatenate([rng.integers(0, 256, (2, h * w)).astype(np.float64), rng.normal(0, 1, (2, 24))], axis=1)
    with torch.no_grad():
        out = net(torch.from_numpy(inp)).numpy()
    ref = pnp.forward_conv(inp, p, h, w)
    assert out.shape == (2, 9)
    assert np.max(np.abs(out - ref)) <= 1e-9 * max(1.0, np.max(np.abs(ref)))
    bi = ini.BatchInitializer(device="cpu", variant="conv")
    assert isinstance(bi.net, ini.PlannerNetConv)


def test_gemm_convolution_equals_direct_convolution():
    """the backbone's im2col + GEMM path (matrix cores on the GPU) computes nn.Conv2d's sums"""
    torch.manual_seed(0)
    for cin, cout, k, s_, pad in ((1, 64, 7, 2, 3), (64, 128, 3, 2, 1), (128, 128, 3, 1, 1), (64, 128, 1, 2, 0)):
        conv = torch.nn.Conv2d(cin, cout, k, s_, pad, bias=False).double()
        x = torch.randn(2, cin, 30, 41, dtype=torch.float64)
        ref = conv(x)
        kh = conv.kernel_size[0]
        cols = x[:, :, ::s_, ::s_].reshape(2, cin, -1) if (kh == 1 and pad == 0) else \
            torch.nn.functional.unfold(x, conv.kernel_size, padding=conv.padding, stride=conv.stride)
        y = torch.matmul(conv.weight.reshape(cout, -1), cols).reshape(ref.shape)
        assert torch.allclose(y, ref, rtol=1e-12, atol=1e-12)
    # and the module-level switch leaves CPU tensors on nn.Conv2d
    assert ini.CONV_IMPL == "gemm"
    c = torch.nn.Conv2d(1, 4, 3, 1, 1, bias=False)
    xx = torch.randn(1, 1, 8, 8)
    assert torch.equal(ini._conv2d(c, xx), c(xx))


def test_raycast_depth_image_of_a_pillar():
    """one 1 m pillar 5 m ahead: it covers the image centre at depth 4.5 m, the ground shows below the horizon, the sky
    (no hit) saturates; scaling to uint8 by the maximum like record_planner.py:16-18"""
    img = ini.raycast_depth([(5.0, 0.0, 1.0, 1.0, 6.0)], eye=(0.0, 0.0, 2.0), max_range=20.0)
    assert img.shape == (ini.IMG_HEIGHT, ini.IMG_WIDTH) and img.dtype == np.uint8
    centre = img[ini.IMG_HEIGHT // 2, ini.IMG_WIDTH // 2]
    assert abs(int(centre) - round(4.5 / 20.0 * 255)) <= 1
    assert img[5, 5] == 255                                   # sky, top-left corner
    assert img[-1, ini.IMG_WIDTH // 2] < 60                   # ground just ahead (2 m below the camera)
    left_edge = img[ini.IMG_HEIGHT // 2, 10]
    assert left_edge == 255                                   # beside the pillar, at the horizon: nothing within range
    # a floating box appears where it should
    img2 = ini.raycast_depth([], [(4.0, 1.0, 3.0, 0.5, 0.5, 0.5)], eye=(0.0, 0.0, 2.0))
    ys, xs = np.nonzero(img2[:ini.IMG_HEIGHT // 2] < 255)     # above the horizon only the box can be hit
    assert len(ys) and xs.mean() < ini.IMG_WIDTH / 2          # the box sits up and to the left (y = +1 m)
    assert abs(int(img2[ys[0], xs[0]]) - round(3.75 / 20.0 * 255)) <= 2
