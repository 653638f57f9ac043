"""
Initializer network of neo-planner on PyTorch-ROCm: the learned warm start
that `NeoPlanner.enhanced_traj_plan` feeds to the optimiser.

reference                                                        here
  nn_trainer/nn_trainer.py:109-155  PlannerNet                    PlannerNet (same sub-module names, so a
                                                                  reference `planner_net.pth` state_dict loads)
  nn_trainer/nn_trainer.py:52-59    process_input_np              process_input_np
  traj_planner/record_planner.py:13-58 form_nn_input              form_nn_input (own quaternion helper)
  traj_planner/nn_planner.py:20-134 NNPlanner (onnxruntime)       NNPlanner (torch forward on the GPU)
  traj_planner/neo_planner.py:10-51 NeoPlanner                    NeoPlanner(MinJerkPlanner)

The reference runs the exported ONNX graph through onnxruntime's CUDA provider; here the same network
runs as a torch module on ROCm (convolutions: MIOpen; dense layers: hipBLASLt/rocBLAS on MFMA).  The
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
        idt = x if self.downsample is None else self.downsample(x)
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
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
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
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

    def __init__(self, net=None, device=None, T_min=0.5, T_max=5.0):
        self.device = torch.device(device) if device is not None else torch.device(
            "cuda" if torch.cuda.is_available() else "cpu")
        self.net = (net if net is not None else PlannerNet()).to(self.device).eval()
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
