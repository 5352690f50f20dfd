"""`ConditionNet` — the ViPC condition encoder of the Score network (reference: model/scorenet/score.py:13-44).

Runs ONCE per `sample()` call, outside the SDE loop (completion_trainer/Latent_SDE_Trainer.py:150-151), and returns
the pair the loop consumes: `(pts_condition (B, hidden, S), img_condition (B, p_dim))`.

  * point branch (score.py:37-41): Conv1d 3->128, LocalGrouper(128, normalize='center') into `patch_size` groups,
    Conv1d 128->hidden — on the HIP kernels shared with the Compressor's front end (FPS, kNN, grouping, bf16 MFMA
    PreExtraction).  Pinned against the reference by tests/golden/condition_net_pts.npz.
  * image branch (score.py:33-36): the first six children of torchvision's resnet18, global max pool, Linear
    128->p_dim.  torchvision is not in this image, so the trunk's parameter tree is laid out here under the names
    its state_dict has upstream (`resnet.0` conv1, `resnet.1` bn1, `resnet.4.{0,1}` layer1, `resnet.5.{0,1}` layer2
    with `downsample.{0,1}` on its first block); the convolutions run as im2col + the fp32 HIP GEMM (BatchNorm folded,
    deterministic); PARITY UNPINNED against torchvision itself (tests compare with an independent fp32 restatement).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .compressor import LocalGrouper, run_grouper
from .layers import _Holder, conv_w


class _BasicBlock(_Holder):
    """torchvision BasicBlock's parameter tree (conv1, bn1, relu, conv2, bn2, downsample), registration order kept."""

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride


def _make_layer(inplanes, planes, blocks, stride):
    down = None
    if stride != 1 or inplanes != planes:
        down = nn.Sequential(nn.Conv2d(inplanes, planes, 1, stride, bias=False), nn.BatchNorm2d(planes))
    return nn.Sequential(_BasicBlock(inplanes, planes, stride, down), *[_BasicBlock(planes, planes) for _ in range(1, blocks)])


def resnet18_trunk():
    """`nn.Sequential(*list(resnet18().children())[:-4])` (score.py:25-26): conv1, bn1, relu, maxpool, layer1, layer2.
    The whole resnet18 is constructed (layer3, layer4, fc too) and initialised the way torchvision does — kaiming-normal
    fan_out convs, unit BatchNorms — so that the global RNG advances as it does upstream; the tail is then dropped."""
    full = nn.Sequential()
    full.add_module("conv1", nn.Conv2d(3, 64, 7, 2, 3, bias=False))
    full.add_module("bn1", nn.BatchNorm2d(64))
    full.add_module("relu", nn.ReLU(inplace=True))
    full.add_module("maxpool", nn.MaxPool2d(3, 2, 1))
    full.add_module("layer1", _make_layer(64, 64, 2, 1))
    full.add_module("layer2", _make_layer(64, 128, 2, 2))
    full.add_module("layer3", _make_layer(128, 256, 2, 2))
    full.add_module("layer4", _make_layer(256, 512, 2, 2))
    full.add_module("avgpool", nn.AdaptiveAvgPool2d((1, 1)))
    full.add_module("fc", nn.Linear(512, 1000))
    for m in full.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        elif isinstance(m, nn.BatchNorm2d):
            nn.init.constant_(m.weight, 1)
            nn.init.constant_(m.bias, 0)
    return nn.Sequential(*list(full.children())[:-4])


class ConditionNet(nn.Module):
    def __init__(self, hidden_size, p_dim, patch_size=16, img_condition=True, pt_condition=True):
        super().__init__()
        self.hidden_size, self.img_condition, self.pt_condition, self.patch_size = hidden_size, img_condition, pt_condition, patch_size
        if pt_condition:
            self.pc_conv_in = nn.Conv1d(3, 128, 1)
            self.group = LocalGrouper(128, True, normalize="center")
            self.pc_conv_out = nn.Conv1d(128, hidden_size, 1)
        if img_condition:
            self.resnet = resnet18_trunk()
            self.ln = nn.Linear(128, p_dim)
        self.conv_out = nn.Conv1d(hidden_size, hidden_size, 1)        # in the reference's state_dict, unused by its forward
        self._pack, self._pack_key = None, None

    def _device(self):
        return self.conv_out.weight.device

    def _packed(self):
        key = tuple((p.data_ptr(), p._version) for p in self.parameters()) + tuple((b.data_ptr(), b._version) for b in self.buffers())
        if self._pack is None or key != self._pack_key:
            self._pack, self._pack_key = ({"group": self.group.pack()} if self.pt_condition else {}), key
        return self._pack

    # ------------------------------------------------------------------ branches
    def points_branch(self, pts):
        """pts (B, n, 3) -> pts_condition (B, hidden, S) channels-first, as the reference returns it."""
        dev = self._device()
        pts = pts.to(dev, torch.float32).contiguous()
        B, n, _ = pts.shape
        f32 = lambda t: t.detach().float().contiguous()
        feat = ops.sgemm(pts.view(B * n, 3), f32(conv_w(self.pc_conv_in)), f32(self.pc_conv_in.bias))   # Conv1d 3 -> 128
        k = feat.shape[1] // self.patch_size * 2          # score.py:40 `x.shape[1]` is the CHANNEL count (128), not n
        if not 0 < k <= n:
            raise ValueError("ConditionNet: k = 128 // patch_size * 2 = %d neighbours of a %d-point cloud" % (k, n))
        _, tok, _, _ = run_grouper(self._packed()["group"], pts, feat.view(B, n, -1), self.patch_size, k)
        out = ops.sgemm(tok, f32(conv_w(self.pc_conv_out)), f32(self.pc_conv_out.bias))                 # [B*S, hidden]
        return out.view(B, self.patch_size, self.hidden_size).transpose(1, 2)

    def image_branch(self, img):
        """img (B, 3, H, W) -> img_condition (B, p_dim): resnet18[:6] (eval-mode BatchNorm), global max pool, Linear.
        Every convolution is im2col (`F.unfold`: a gather) + the fp32 HIP GEMM with the BatchNorm folded into its weights and
        the ReLU into its epilogue — deterministic run to run, unlike the library convolutions."""
        from ._lib import ACT_NONE, ACT_RELU
        r = self.resnet
        h = img.to(self._device(), torch.float32).contiguous()

        def conv_bn(x, conv, bn, relu):
            B, _, H, W = x.shape
            k, st, pd = conv.kernel_size[0], conv.stride[0], conv.padding[0]
            Ho, Wo = (H + 2 * pd - k) // st + 1, (W + 2 * pd - k) // st + 1
            s = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
            w = (conv.weight.detach().float().flatten(1) * s[:, None]).contiguous()
            b = (bn.bias.detach().float() - bn.running_mean.float() * s).contiguous()
            cols = F.unfold(x, k, padding=pd, stride=st).transpose(1, 2).reshape(B * Ho * Wo, -1).contiguous()
            y = ops.sgemm(cols, w, b, act_out=ACT_RELU if relu else ACT_NONE)                   # [B*Ho*Wo, Cout]
            return y.view(B, Ho * Wo, -1).transpose(1, 2).reshape(B, -1, Ho, Wo)

        h = F.max_pool2d(conv_bn(h, r[0], r[1], True), 3, 2, 1)
        for layer in (r[4], r[5]):
            for blk in layer:
                y = conv_bn(conv_bn(h, blk.conv1, blk.bn1, True), blk.conv2, blk.bn2, False)
                if blk.downsample is not None:
                    h = conv_bn(h, blk.downsample[0], blk.downsample[1], False)
                h = F.relu(y + h)
        h = F.adaptive_max_pool2d(h, 1).flatten(1).contiguous()
        return ops.sgemm(h, self.ln.weight.detach().float().contiguous(), self.ln.bias.detach().float().contiguous())

    @torch.no_grad()
    def forward(self, condition):
        """condition: {'img': (B,3,H,W), 'pts': (B,n,3)} (either may be missing) -> (pts_condition | 0., img_condition | 0.)"""
        if self._device().type != "cuda":
            raise RuntimeError("ConditionNet parameters are on %s: the HIP path has no CPU fallback" % self._device())
        x, img = 0., 0.
        if "img" in condition and self.img_condition:
            img = self.image_branch(condition["img"])
        if "pts" in condition and self.pt_condition:
            x = self.points_branch(condition["pts"])
        return x, img
