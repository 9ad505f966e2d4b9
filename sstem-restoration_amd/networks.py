"""SP networks on MI355X: ``IFNet`` (8 live + 8 dead kernel heads, 4 sepconv calls, 2 output channels),
``UNet`` (correction) and ``FusionNet`` (fusion) with the blocks ``DoubleConv / Down / Up / OutConv`` --
same class names, constructors, sub-module names and dataflow as the reference
``sp_scripts_train/networks.py:9-306`` (identical copy in ``sp_scripts_test``).  Convolution runs are
fused native launches (``hipnn``); the local convolutions are the native sepconv op."""
import torch
import torch.nn as nn
import torch.nn.functional as F
import torch.nn.init as init

from hipnn import FusedSequential
from hipnn.fused import run_fused
from libs.sepconv.SeparableConvolution import SeparableConvolution
from libs.sepconv.fused import interp_apply


def _conv3(cin, cout):
    return nn.Conv2d(cin, cout, (3, 3), (1, 1), 1)


class IFNet(nn.Module):
    def __init__(self):
        super(IFNet, self).__init__()
        taps = 51

        self.pool = nn.AvgPool2d(kernel_size=(2, 2), stride=(2, 2))
        self.upsamp = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.relu = nn.ReLU(inplace=False)

        self.conv32 = self._conv_module(6, 32)
        self.conv64 = self._conv_module(32, 64)
        self.conv128 = self._conv_module(64, 128)
        self.conv256 = self._conv_module(128, 256)
        self.conv512 = self._conv_module(256, 512)
        self.conv512x512 = self._conv_module(512, 512)
        self.upsamp512 = self._upsample_module(512, 512)
        self.upconv256 = self._conv_module(512, 256)
        self.upsamp256 = self._upsample_module(256, 256)
        self.upconv128 = self._conv_module(256, 128)
        self.upsamp128 = self._upsample_module(128, 128)
        self.upconv64 = self._conv_module(128, 64)
        self.upsamp64 = self._upsample_module(64, 64)
        # 16 heads are registered (reference :34-66); forward uses _11.._14 and _21.._24 only
        for group in (1, 2):
            for k in range(1, 9):
                setattr(self, "upconv51_%d%d" % (group, k), self._kernel_module(64, taps))

        self.pad = nn.ReplicationPad2d(taps // 2)
        self.separable_conv = SeparableConvolution.apply

        self.apply(self._weight_init)

    def forward(self, x):
        i1 = x[:, :3]
        i2 = x[:, 3:6]

        x = self.conv32(x)
        x = self.pool(x)
        x64 = self.conv64(x)
        x128 = self.conv128(self.pool(x64))
        x256 = self.conv256(self.pool(x128))
        x512 = self.conv512(self.pool(x256))
        x = self.conv512x512(self.pool(x512))

        x = self.upsamp512(x)
        x += x512
        x = self.upconv256(x)
        x = self.upsamp256(x)
        x += x256
        x = self.upconv128(x)
        x = self.upsamp128(x)
        x += x128
        x = self.upconv64(x)
        x = self.upsamp64(x)
        x += x64

        k11h = self.upconv51_11(x)
        k11v = self.upconv51_12(x)
        k12h = self.upconv51_13(x)
        k12v = self.upconv51_14(x)
        k21h = self.upconv51_21(x)
        k21v = self.upconv51_22(x)
        k22h = self.upconv51_23(x)
        k22v = self.upconv51_24(x)

        if not torch.is_grad_enabled():
            # inference: each output channel's pad + 2 local convolutions + add + mean is one launch
            return torch.cat((interp_apply(i1, i2, k11v, k11h, k12v, k12h),
                              interp_apply(i1, i2, k21v, k21h, k22v, k22h)), 1)

        padded_i2 = self.pad(i2).contiguous()
        padded_i1 = self.pad(i1).contiguous()

        # reference :120-126
        y1 = self.separable_conv(padded_i2, k12v, k12h) + self.separable_conv(padded_i1, k11v, k11h)
        y1 = torch.mean(y1, dim=1, keepdim=True)
        y2 = self.separable_conv(padded_i2, k22v, k22h) + self.separable_conv(padded_i1, k21v, k21h)
        y2 = torch.mean(y2, dim=1, keepdim=True)
        return torch.cat((y1, y2), 1)

    def _conv_module(self, cin, cout):
        return FusedSequential(_conv3(cin, cin), self.relu, _conv3(cin, cin), self.relu, _conv3(cin, cout), self.relu)

    def _kernel_module(self, cin, cout):
        return FusedSequential(_conv3(cin, cin), self.relu, _conv3(cin, cin), self.relu, _conv3(cin, cout), self.relu,
                               self.upsamp, _conv3(cout, cout))

    def _upsample_module(self, cin, cout):
        return FusedSequential(self.upsamp, _conv3(cin, cout), self.relu)

    @staticmethod
    def _weight_init(m):
        if isinstance(m, nn.Conv2d):
            init.orthogonal_(m.weight, init.calculate_gain('relu'))


class DoubleConv(nn.Module):
    """(convolution => [BN] => ReLU) * 2   (reference :172-189)"""

    def __init__(self, in_channels, out_channels, mid_channels=None):
        super().__init__()
        if not mid_channels:
            mid_channels = out_channels
        self.double_conv = FusedSequential(
            nn.Conv2d(in_channels, mid_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(mid_channels),
            nn.ReLU(inplace=True),
            nn.Conv2d(mid_channels, out_channels, kernel_size=3, padding=1),
            nn.BatchNorm2d(out_channels),
            nn.ReLU(inplace=True)
        )

    def forward(self, x):
        return self.double_conv(x)


class Down(nn.Module):
    """maxpool then double conv   (reference :192-203)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x):
        return self.maxpool_conv(x)


class Up(nn.Module):
    """upscale, pad to the skip's size, cat([skip, up]), double conv   (reference :206-232)"""

    def __init__(self, in_channels, out_channels, bilinear=True):
        super().__init__()
        if bilinear:
            self.up = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
            self.conv = DoubleConv(in_channels, out_channels, in_channels // 2)
        else:
            self.up = nn.ConvTranspose2d(in_channels, in_channels // 2, kernel_size=2, stride=2)
            self.conv = DoubleConv(in_channels, out_channels)

    def forward(self, x1, x2):
        x1 = self.up(x1)
        diffY = x2.size()[2] - x1.size()[2]
        diffX = x2.size()[3] - x1.size()[3]
        x1 = F.pad(x1, [diffX // 2, diffX - diffX // 2, diffY // 2, diffY - diffY // 2])
        return self.conv(torch.cat([x2, x1], dim=1))   # skip first (reference :231)


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super(OutConv, self).__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        return run_fused([self.conv], x)


class UNet(nn.Module):
    def __init__(self, n_channels, n_classes, bilinear=True):
        super(UNet, self).__init__()
        self.n_channels = n_channels
        self.n_classes = n_classes
        self.bilinear = bilinear

        self.inc = DoubleConv(n_channels, 64)
        self.down1 = Down(64, 128)
        self.down2 = Down(128, 256)
        self.down3 = Down(256, 512)
        factor = 2 if bilinear else 1
        self.down4 = Down(512, 1024 // factor)
        self.up1 = Up(1024, 512 // factor, bilinear)
        self.up2 = Up(512, 256 // factor, bilinear)
        self.up3 = Up(256, 128 // factor, bilinear)
        self.up4 = Up(128, 64, bilinear)
        self.outc = OutConv(64, n_classes)

    def forward(self, x):
        x1 = self.inc(x)
        x2 = self.down1(x1)
        x3 = self.down2(x2)
        x4 = self.down3(x3)
        x5 = self.down4(x4)
        x = self.up1(x5, x4)
        x = self.up2(x, x3)
        x = self.up3(x, x2)
        x = self.up4(x, x1)
        return self.outc(x)


class FusionNet(UNet):
    """Same network; the two inputs are added first (reference :275-306)."""

    def forward(self, x_in1, x_in2):
        return super().forward(torch.add(x_in1, x_in2))
