"""
oracle/plannernet_np.py -- TEST INFRASTRUCTURE ONLY.

fp64 NumPy forward of the initializer network, layer by layer from the reference's definition
(nn_trainer/nn_trainer.py:109-155: torchvision ResNet-18 with a 1-channel 7x7 stem and a 24-wide fc,
motion MLP 24-48-24-24-24, head MLP 48-48-96-96-9, LeakyReLU(0.01)), taking a torch-style
state_dict of NumPy arrays.  Parity status: UNPINNED against the reference -- its trained weights
(saved_net/planner_net.{pth,onnx}) are not in the tree and torchvision/onnxruntime are not in this
image -- so this file checks architecture and data flow of neo_planner_amd.initializer only.
"""
import numpy as np


def conv2d(x, w, stride, pad):
    """x [N,C,H,W], w [O,C,kh,kw], no bias"""
    N, C, H, W = x.shape
    O, _, kh, kw = w.shape
    xp = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)))
    Ho = (H + 2 * pad - kh) // stride + 1
    Wo = (W + 2 * pad - kw) // stride + 1
    s = xp.strides
    win = np.lib.stride_tricks.as_strided(xp, (N, C, Ho, Wo, kh, kw),
                                          (s[0], s[1], s[2] * stride, s[3] * stride, s[2], s[3]))
    return np.einsum("nchwij,ocij->nohw", win, w, optimize=True)


def batchnorm(x, p, prefix, eps=1e-5):
    g, b = p[prefix + ".weight"], p[prefix + ".bias"]
    m, v = p[prefix + ".running_mean"], p[prefix + ".running_var"]
    return (x - m[None, :, None, None]) / np.sqrt(v[None, :, None, None] + eps) * g[None, :, None, None] + b[None, :, None, None]


def maxpool3s2p1(x):
    N, C, H, W = x.shape
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1), (1, 1)), constant_values=-np.inf)
    Ho, Wo = (H + 2 - 3) // 2 + 1, (W + 2 - 3) // 2 + 1
    s = xp.strides
    win = np.lib.stride_tricks.as_strided(xp, (N, C, Ho, Wo, 3, 3), (s[0], s[1], s[2] * 2, s[3] * 2, s[2], s[3]))
    return win.max(axis=(4, 5))


def relu(x):
    return np.maximum(x, 0.0)


def leaky(x):
    return np.where(x > 0, x, 0.01 * x)


def linear(x, p, prefix):
    return x @ p[prefix + ".weight"].T + p[prefix + ".bias"]


def basic_block(x, p, prefix, stride, down):
    idt = x
    if down:
        idt = batchnorm(conv2d(x, p[prefix + ".downsample.0.weight"], stride, 0), p, prefix + ".downsample.1")
    y = relu(batchnorm(conv2d(x, p[prefix + ".conv1.weight"], stride, 1), p, prefix + ".bn1"))
    y = batchnorm(conv2d(y, p[prefix + ".conv2.weight"], 1, 1), p, prefix + ".bn2")
    return relu(y + idt)


def image_features(img, p):
    b = "img_backbone"
    x = maxpool3s2p1(relu(batchnorm(conv2d(img, p[b + ".conv1.weight"], 2, 3), p, b + ".bn1")))
    for layer, stride in (("layer1", 1), ("layer2", 2), ("layer3", 2), ("layer4", 2)):
        x = basic_block(x, p, f"{b}.{layer}.0", stride, down=(layer != "layer1"))
        x = basic_block(x, p, f"{b}.{layer}.1", 1, down=False)
    return linear(x.mean(axis=(2, 3)), p, b + ".fc")


def head(img_feature, motion, p):
    m = motion
    for i in (0, 2, 4):
        m = leaky(linear(m, p, f"motion_backbone.{i}"))
    m = linear(m, p, "motion_backbone.6")
    x = np.concatenate([np.broadcast_to(img_feature, (m.shape[0], img_feature.shape[1])), m], axis=1)
    for i in (0, 2, 4):
        x = leaky(linear(x, p, f"mlp.{i}"))
    return linear(x, p, "mlp.6")


def forward(inp, p, img_height, img_width):
    """the reference's PlannerNet.forward: inp [N, H*W + 24]"""
    hw = img_height * img_width
    img = inp[:, :hw].reshape(-1, 1, img_height, img_width)
    return head(image_features(img, p), inp[:, hw:], p)


# ---- the Conv1d variant (nn_trainer/nn_trainer_conv.py:107-159)
def conv1d_k3p1(x, p, prefix):
    """x [N, C, L], kernel 3, stride 1, padding 1, with bias"""
    w, b = p[prefix + ".weight"], p[prefix + ".bias"]          # [O, C, 3], [O]
    xp = np.pad(x, ((0, 0), (0, 0), (1, 1)))
    L = x.shape[2]
    win = np.stack([xp[:, :, k:k + L] for k in range(3)], axis=-1)     # [N, C, L, 3]
    return np.einsum("nclk,ock->nol", win, w, optimize=True) + b[None, :, None]


def _conv_stack(x, p, prefix):
    """Sequential(Conv1d, LeakyReLU, Conv1d, LeakyReLU, Conv1d, LeakyReLU, Flatten, Linear): indices 0, 2, 4, 7"""
    y = x[:, None, :]
    for i in (0, 2, 4):
        y = leaky(conv1d_k3p1(y, p, f"{prefix}.{i}"))
    return linear(y.reshape(y.shape[0], -1), p, f"{prefix}.7")


def head_conv(img_feature, motion, p):
    m = _conv_stack(motion, p, "motion_backbone")
    x = np.concatenate([np.broadcast_to(img_feature, (m.shape[0], img_feature.shape[1])), m], axis=1)
    return _conv_stack(x, p, "mlp")


def forward_conv(inp, p, img_height, img_width):
    hw = img_height * img_width
    img = inp[:, :hw].reshape(-1, 1, img_height, img_width)
    return head_conv(image_features(img, p), inp[:, hw:], p)
