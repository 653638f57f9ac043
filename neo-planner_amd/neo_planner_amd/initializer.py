"""
Initializer network of neo-planner on PyTorch-ROCm: the learned warm start
that `NeoPlanner.enhanced_traj_plan` feeds to the optimiser.

reference                                                        here
  nn_trainer/nn_trainer.py:109-155  PlannerNet                    PlannerNet (same sub-module names, so a
                                                                  reference `planner_net.pth` state_dict loads)
  nn_trainer/nn_trainer_conv.py:107-159 PlannerNet (Conv1d heads) PlannerNetConv (same sub-module names)
  nn_trainer/nn_trainer.py:52-59    process_input_np              process_input_np
  traj_planner/record_planner.py:13-58 form_nn_input              form_nn_input (own quaternion helper)
  traj_planner/nn_planner.py:20-134 NNPlanner (onnxruntime)       NNPlanner (torch forward on the GPU)
  traj_planner/neo_planner.py:10-51 NeoPlanner                    NeoPlanner(MinJerkPlanner)

The reference runs the exported ONNX graph through onnxruntime's CUDA provider; here the same network
runs as a torch module on ROCm.  The backbone's convolutions run as im2col + GEMM (`conv_impl="gemm"`: F.unfold, then
one fp32 GEMM per layer on hipBLASLt/rocBLAS, i.e. on the matrix cores) because MIOpen picks its naive direct kernel
for these fp32 NCHW shapes on gfx950 (profiles/r01_cfg3_*: 63 % of the warm start's GPU time); `conv_impl="miopen"`
keeps nn.Conv2d.  Dense layers: hipBLASLt/rocBLAS on MFMA.  The
trained weights are not part of the reference tree (.MISSING_LARGE_BLOBS): parity of the *numbers* is
unpinned, parity of the *architecture and data flow* is tested against an fp64 NumPy forward of the
same weights (tests/test_initializer.py).  torchvision is not available in this image: the ResNet-18
below is written out with torchvision's parameter names.
"""
import numpy as np
import torch
import torch.nn as nn

from .planner import MinJerkPlanner

IMG_WIDTH = 640            # nn_trainer.py:19-22
IMG_HEIGHT = 480
MOTION_INPUT_SIZE = 24
OUTPUT_SIZE = 9
IMG_FEATURE_SIZE = 24
MOTION_FEATURE_SIZE = 24


# --------------------------------------------------------------------------- quaternion (pyquaternion subset)
class Quat:
    """unit quaternion with the members the reference uses from pyquaternion: `rotate`, `inverse`,
    `rotation_matrix` (record_planner.py:21-42, nn_planner.py:128-132)"""

    def __init__(self, w=1.0, x=0.0, y=0.0, z=0.0):
        q = np.array([w, x, y, z], dtype=np.float64)
        self.q = q / np.linalg.norm(q)

    @classmethod
    def from_yaw(cls, yaw):
        return cls(np.cos(yaw / 2), 0.0, 0.0, np.sin(yaw / 2))

    @property
    def rotation_matrix(self):
        w, x, y, z = self.q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    @property
    def inverse(self):
        w, x, y, z = self.q
        return Quat(w, -x, -y, -z)

    def rotate(self, v):
        return self.rotation_matrix @ np.asarray(v, dtype=np.float64)


class DroneState:
    """ros_node/traj_planner_node.py:49-55"""

    def __init__(self):
        self.global_pos = np.zeros(3)
        self.global_vel = np.zeros(3)
        self.local_vel = np.zeros(3)
        self.attitude = Quat()
        self.yaw = 0.0


# --------------------------------------------------------------------------- input glue
def process_input_np(depth_img, motion_info):
    """nn_trainer.py:52-59: flatten the image, append the motion vector, float32"""
    return np.concatenate((depth_img.reshape(-1).astype(np.float32), motion_info.astype(np.float32)))


def form_nn_input(depth_img, drone_state, des_pos_z, plan_init_state, target_state):
    """record_planner.py:13-58: depth image scaled to uint8 by its maximum; 24-d motion vector
    [local velocity (3), attitude matrix row-major (9), plan start pos/vel and target pos/vel in the
    body frame (4 x 3)]"""
    depth_norm = (depth_img / np.max(depth_img) * 255).astype(np.uint8)
    q = drone_state.attitude
    start = np.zeros((2, 3))
    start[0, :2] = plan_init_state.global_pos[:2]
    start[0, 2] = des_pos_z
    start[1, :2] = plan_init_state.global_vel[:2]
    goal = np.zeros((2, 3))
    goal[:, :2] = target_state
    goal[0, 2] = des_pos_z
    to_body = q.inverse
    motion = np.concatenate((drone_state.local_vel,
                             q.rotation_matrix.reshape(-1),
                             to_body.rotate(start[0] - drone_state.global_pos),
                             to_body.rotate(start[1] - drone_state.global_vel),
                             to_body.rotate(goal[0] - drone_state.global_pos),
                             to_body.rotate(goal[1] - drone_state.global_vel)), axis=0)
    return depth_norm, motion


# --------------------------------------------------------------------------- network
CONV_IMPL = "gemm"      # "gemm": im2col + one GEMM per convolution (matrix cores); "miopen": nn.Conv2d as is


def _conv2d(conv, x):
    """nn.Conv2d forward.  With CONV_IMPL == "gemm" on a GPU tensor: F.unfold (im2col) and a [Cout, Cin*kh*kw] x
    [Cin*kh*kw, L] GEMM per image -- the same sums as the direct convolution in another order (fp32 round-off apart)."""
    if CONV_IMPL != "gemm" or not x.is_cuda or conv.groups != 1:
        return conv(x)
    kh, kw = conv.kernel_size
    N, _, H, W = x.shape
    Ho = (H + 2 * conv.padding[0] - conv.dilation[0] * (kh - 1) - 1) // conv.stride[0] + 1
    Wo = (W + 2 * conv.padding[1] - conv.dilation[1] * (kw - 1) - 1) // conv.stride[1] + 1
    if kh == 1 and kw == 1 and conv.padding == (0, 0):
        cols = x[:, :, ::conv.stride[0], ::conv.stride[1]].reshape(N, x.shape[1], -1)
    else:
        cols = torch.nn.functional.unfold(x, (kh, kw), dilation=conv.dilation, padding=conv.padding, stride=conv.stride)
    y = torch.matmul(conv.weight.reshape(conv.out_channels, -1), cols)           # [N, Cout, L]
    if conv.bias is not None:
        y = y + conv.bias[None, :, None]
    return y.reshape(N, conv.out_channels, Ho, Wo)


class _BasicBlock(nn.Module):
    def __init__(self, cin, cout, stride):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, cout, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(cout)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(cout, cout, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(cout)
        self.downsample = None
        if stride != 1 or cin != cout:
            self.downsample = nn.Sequential(nn.Conv2d(cin, cout, 1, stride, bias=False), nn.BatchNorm2d(cout))

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample[1](_conv2d(self.downsample[0], x))
        y = self.relu(self.bn1(_conv2d(self.conv1, x)))
        y = self.bn2(_conv2d(self.conv2, y))
        return self.relu(y + idt)


class ResNet18OneChannel(nn.Module):
    """torchvision.models.resnet18 with the reference's two edits (nn_trainer.py:119-122): a 1-channel
    stem and a `feature_size`-wide fc.  Parameter names follow torchvision."""

    def __init__(self, feature_size=IMG_FEATURE_SIZE):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = nn.Sequential(_BasicBlock(64, 64, 1), _BasicBlock(64, 64, 1))
        self.layer2 = nn.Sequential(_BasicBlock(64, 128, 2), _BasicBlock(128, 128, 1))
        self.layer3 = nn.Sequential(_BasicBlock(128, 256, 2), _BasicBlock(256, 256, 1))
        self.layer4 = nn.Sequential(_BasicBlock(256, 512, 2), _BasicBlock(512, 512, 1))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512, feature_size)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(_conv2d(self.conv1, x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


class PlannerNet(nn.Module):
    """nn_trainer.py:109-155.  `forward(input)` is the reference's signature ([N, H*W + 24] float32 ->
    [N, 9]); `image_features` / `head` split it so that a batch of trajectories sharing one depth image
    runs the convolutional backbone once."""

    def __init__(self, img_height=IMG_HEIGHT, img_width=IMG_WIDTH):
        super().__init__()
        self.img_height, self.img_width = img_height, img_width
        self.img_backbone = ResNet18OneChannel(IMG_FEATURE_SIZE)
        self.motion_backbone = nn.Sequential(
            nn.Linear(MOTION_INPUT_SIZE, 48), nn.LeakyReLU(),
            nn.Linear(48, 24), nn.LeakyReLU(),
            nn.Linear(24, 24), nn.LeakyReLU(),
            nn.Linear(24, MOTION_FEATURE_SIZE))
        self.mlp = nn.Sequential(
            nn.Linear(IMG_FEATURE_SIZE + MOTION_FEATURE_SIZE, 48), nn.LeakyReLU(),
            nn.Linear(48, 96), nn.LeakyReLU(),
            nn.Linear(96, 96), nn.LeakyReLU(),
            nn.Linear(96, OUTPUT_SIZE))

    def image_features(self, img):
        """img [N, 1, H, W] -> [N, 24]"""
        return self.img_backbone(img)

    def head(self, img_feature, motion):
        """img_feature [N or 1, 24], motion [N, 24] -> [N, 9]: 20 448 MAC per trajectory of dense work"""
        if img_feature.shape[0] != motion.shape[0]:
            img_feature = img_feature.expand(motion.shape[0], -1)
        return self.mlp(torch.cat([img_feature, self.motion_backbone(motion)], dim=1))

    def forward(self, input):
        hw = self.img_width * self.img_height
        img = input[:, :hw].reshape(-1, 1, self.img_height, self.img_width)
        return self.head(self.image_features(img), input[:, hw:])


class PlannerNetConv(PlannerNet):
    """nn_trainer_conv.py:107-159: the variant whose motion branch and fusion head are Conv1d stacks over the feature
    vector treated as a 1-channel sequence (kernel 3, padding 1, 1 -> 16 -> 32 -> 64 channels, LeakyReLU), flattened
    into one Linear each.  Same sub-module names and Sequential indices as the reference, so its state_dict loads.
    Per trajectory: motion branch 24 * (3*16 + 48*32 + 96*64) + 1536*24 MAC, head 48 * (3*16 + 48*32 + 96*64) + 3072*9."""

    def __init__(self, img_height=IMG_HEIGHT, img_width=IMG_WIDTH):
        super().__init__(img_height, img_width)

        def stack(length, out):
            return nn.Sequential(
                nn.Conv1d(1, 16, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(),
                nn.Conv1d(16, 32, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(),
                nn.Conv1d(32, 64, kernel_size=3, stride=1, padding=1), nn.LeakyReLU(),
                nn.Flatten(), nn.Linear(64 * length, out))
        self.motion_backbone = stack(MOTION_INPUT_SIZE, MOTION_FEATURE_SIZE)
        self.mlp = stack(IMG_FEATURE_SIZE + MOTION_FEATURE_SIZE, OUTPUT_SIZE)

    def head(self, img_feature, motion):
        if img_feature.shape[0] != motion.shape[0]:
            img_feature = img_feature.expand(motion.shape[0], -1)
        mf = self.motion_backbone(motion.unsqueeze(1))                 # (:149-152) channel dimension added
        return self.mlp(torch.cat([img_feature, mf], dim=1).unsqueeze(1))


def raycast_depth(pillars, canopy=(), eye=(1.5, 0.0, 2.0), yaw=0.0, hfov_deg=87.0, max_range=20.0,
                  height=IMG_HEIGHT, width=IMG_WIDTH):
    """Synthetic depth image of a forest scene ("ray-cast of the pillars, 480 x 640"): a pinhole camera
    at `eye` looking along +x rotated by `yaw`, horizontal field of view `hfov_deg` (a RealSense-like 87 deg), depth =
    distance along the optical axis to the nearest box or the ground plane z = 0, capped at `max_range`; returned as
    uint8 scaled by its maximum, exactly what form_nn_input hands to the network (record_planner.py:16-18).
    pillars: (cx, cy, sx, sy, sz) standing on the ground; canopy: (cx, cy, cz, sx, sy, sz)."""
    boxes = [((cx - sx / 2, cy - sy / 2, 0.0), (cx + sx / 2, cy + sy / 2, sz)) for (cx, cy, sx, sy, sz) in pillars]
    boxes += [((cx - sx / 2, cy - sy / 2, cz - sz / 2), (cx + sx / 2, cy + sy / 2, cz + sz / 2))
              for (cx, cy, cz, sx, sy, sz) in canopy]
    f = (width / 2) / np.tan(np.radians(hfov_deg) / 2)
    u = (np.arange(width) - (width - 1) / 2) / f
    v = (np.arange(height) - (height - 1) / 2) / f
    # camera frame: x forward, y left, z up
    d = np.stack(np.broadcast_arrays(np.ones((height, width)), -u[None, :], -v[:, None]), axis=-1)
    c, s_ = np.cos(yaw), np.sin(yaw)
    R = np.array([[c, -s_, 0.0], [s_, c, 0.0], [0.0, 0.0, 1.0]])
    dirs = d @ R.T                                                      # world directions, forward component = 1
    eye = np.asarray(eye, dtype=np.float64)
    depth = np.full((height, width), max_range)
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(dirs[..., 2] < 0, -eye[2] / dirs[..., 2], np.inf)          # ground plane
        depth = np.minimum(depth, tg)
        inv = 1.0 / dirs
        for lo, hi in boxes:
            t0 = (np.asarray(lo) - eye) * inv
            t1 = (np.asarray(hi) - eye) * inv
            tn = np.minimum(t0, t1).max(axis=-1)
            tf = np.maximum(t0, t1).min(axis=-1)
            hit = (tf >= np.maximum(tn, 0.0))
            depth = np.where(hit, np.minimum(depth, np.maximum(tn, 0.0)), depth)
    depth = np.clip(depth, 0.0, max_range)
    return (depth / max(depth.max(), 1e-9) * 255).astype(np.uint8)


def split_output(out, M=3, nn_output_D=3):
    """nn_planner.py:104-105: 9 outputs -> (M-1) body-frame 3-D waypoints (column major) and M durations"""
    out = np.asarray(out)
    wpts_local = out[:nn_output_D * (M - 1)].reshape(M - 1, nn_output_D).T
    return wpts_local, out[nn_output_D * (M - 1):]


class NNPlanner:
    """nn_planner.py:20-134 with a torch module in place of the onnxruntime session"""

    def __init__(self, des_pos_z=2.0, net=None, device=None, state_dict_path=None):
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.net = net if net is not None else PlannerNet()
        if state_dict_path is not None:
            self.net.load_state_dict(torch.load(state_dict_path, map_location="cpu"))
        self.net = self.net.to(self.device).eval()
        self.M, self.s, self.D, self.nn_output_D = 3, 3, 2, 3          # nn_planner.py:57-66
        self.head_state = np.zeros((self.s, self.D))
        self.tail_state = np.zeros((self.s, self.D))
        self.des_pos_z = des_pos_z

    def nn_traj_plan(self, depth_img, drone_state, plan_init_state, target_state):
        depth_norm, motion = form_nn_input(depth_img, drone_state, self.des_pos_z, plan_init_state, target_state)
        self.drone_state = drone_state
        self.head_state[0, :self.D] = plan_init_state.global_pos[:2]
        self.head_state[1, :self.D] = plan_init_state.global_vel[:2]
        self.tail_state[0, :self.D] = target_state[0, :2]
        self.tail_state[1, :self.D] = target_state[1, :2]
        self.predict(depth_norm, motion)

    def predict(self, depth_image_norm, motion_info):
        """nn_planner.py:87-108 (`onnx_predict`)"""
        inp = torch.from_numpy(np.array([process_input_np(depth_image_norm, motion_info)])).to(self.device)
        with torch.no_grad():
            out = self.net(inp)[0].float().cpu().numpy()
        wpts_local, self.ts = split_output(out, self.M, self.nn_output_D)
        self.int_wpts = self.get_wpts_world(wpts_local)[:self.D, :]

    onnx_predict = predict

    def get_wpts_world(self, int_wpts):
        """nn_planner.py:123-134: body frame -> world frame"""
        out = np.zeros((self.nn_output_D, self.M - 1))
        for i in range(self.M - 1):
            out[:, i] = self.drone_state.attitude.rotate(int_wpts[:, i]) + self.drone_state.global_pos
        return out


class NeoPlanner(MinJerkPlanner):
    """neo_planner.py:10-51: network output as the warm start of the optimiser"""

    def __init__(self, planner_config, nn_planner=None, **kw):
        super().__init__(planner_config, **kw)
        self.nn_planner = nn_planner if nn_planner is not None else NNPlanner(getattr(planner_config, "des_pos_z", 2.0))

    def enhanced_traj_plan(self, map, depth_img, drone_state, plan_init_state, target_state):
        self.nn_planner.nn_traj_plan(depth_img, drone_state, plan_init_state, target_state)
        start_2d = np.array([plan_init_state.global_pos[:2], plan_init_state.global_vel[:2]])
        self.warm_start_plan(map, start_2d, target_state, self.nn_planner.int_wpts, self.nn_planner.ts)


class BatchInitializer:
    """cfg3: warm starts for B trajectories that share one depth image.  The backbone runs once per
    scene; the dense layers run as [B, 48] x ... GEMMs."""

    def __init__(self, net=None, device=None, T_min=0.5, T_max=5.0, variant="mlp"):
        """variant: "mlp" (nn_trainer.py) or "conv" (nn_trainer_conv.py) when no `net` is given"""
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        if net is None:
            net = PlannerNet() if variant == "mlp" else PlannerNetConv()
        self.net = net.to(self.device).eval()
        self.T_min, self.T_max = T_min, T_max

    @torch.no_grad()
    def scene_feature(self, depth_norm):
        img = torch.as_tensor(np.asarray(depth_norm), dtype=torch.float32, device=self.device)
        return self.net.image_features(img.reshape(1, 1, *img.shape[-2:]))

    @torch.no_grad()
    def warm_start(self, scene_feature, motion, attitude_R, global_pos, clamp_ts=True):
        """motion [B,24], attitude_R [B,3,3] (body->world), global_pos [B,3]  ->
        int_wpts [B,2,2] (world, z dropped), ts [B,3].  An untrained or extrapolating network can
        emit durations outside (T_min, T_max), where the reference's map_T2tau fails; `clamp_ts`
        keeps them inside."""
        motion = torch.as_tensor(motion, dtype=torch.float32, device=self.device)
        out = self.net.head(scene_feature, motion).double()
        R = torch.as_tensor(attitude_R, dtype=torch.float64, device=self.device)
        p = torch.as_tensor(global_pos, dtype=torch.float64, device=self.device)
        local = out[:, :6].reshape(-1, 2, 3)                       # [B, waypoint, xyz]
        world = torch.einsum("bij,bwj->bwi", R, local) + p[:, None, :]
        ts = out[:, 6:]
        if clamp_ts:
            eps = 1e-3 * (self.T_max - self.T_min)
            ts = ts.clamp(self.T_min + eps, self.T_max - eps)
        return world[:, :, :2].transpose(1, 2).contiguous(), ts.contiguous()
