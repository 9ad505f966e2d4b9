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
import hipnn.functional as HF
from hipnn.fused import run_fused
from libs.sepconv.SeparableConvolution import SeparableConvolution
from libs.sepconv.fused import (coef_to_blocked, interp_apply, interp_apply_gray, interp_apply_gray_blocked,
                                interp_apply_gray_blocked_supported, interp_apply_gray_supported)


def _conv3(cin, cout):
    return nn.Conv2d(cin, cout, (3, 3), (1, 1), 1)


class IFNet(nn.Module):
    """Encoder 6 -> 32 -> 64 -> 128 -> 256 -> 512 -> 512 (average pooling between levels), decoder back up to 64 channels at
    half resolution with additive skips, then kernel heads at full resolution: two output channels, each the channel mean of
    two 51-tap local convolutions (one per input frame).  Built from tables; attribute names and registration order are the
    reference's (:9-170), incl. its eight unused heads, so its checkpoints load with strict=True."""
    ENCODER = ((32, 6), (64, 32), (128, 64), (256, 128), (512, 256))          # name suffix = width, input channels
    DECODER = (512, 256, 128, 64)                                              # upsamp<w> (+ skip), then upconv<w/2>

    def __init__(self):
        super().__init__()
        taps = 51
        self.pool = nn.AvgPool2d(kernel_size=(2, 2), stride=(2, 2))
        self.upsamp = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.relu = nn.ReLU(inplace=False)
        for w, cin in self.ENCODER:
            setattr(self, "conv%d" % w, self._conv_module(cin, w))
        self.conv512x512 = self._conv_module(512, 512)
        for w in self.DECODER:
            setattr(self, "upsamp%d" % w, self._upsample_module(w, w))
            if w > 64:
                setattr(self, "upconv%d" % (w // 2), self._conv_module(w, w // 2))
        # 16 heads are registered (reference :34-66); forward uses _g1.._g4 of each output channel g only
        for group in (1, 2):
            for k in range(1, 9):
                setattr(self, "upconv51_%d%d" % (group, k), self._kernel_module(64, taps))
        self.pad = nn.ReplicationPad2d(taps // 2)
        self.separable_conv = SeparableConvolution.apply
        self.apply(self._weight_init)

    def forward(self, x):
        return self._interpolate(x, None)

    def interpolate_gray(self, frame1, frame2):
        """``forward`` for callers that hold the two grayscale planes [B,1,H,W] and would build the network input by
        replicating each x3 (reference test_fusion.py:105-106, main_fusion.py:210-211).  Same result, bit for bit; at
        inference the local convolutions run on the planes themselves (no channel comparison on the device)."""
        B, _, H, W = frame1.shape
        x = torch.cat((frame1.expand(B, 3, H, W), frame2.expand(B, 3, H, W)), 1)
        return self._interpolate(x, (frame1, frame2))

    def _interpolate(self, x, gray):
        i1, i2 = x[:, :3], x[:, 3:6]
        # (`pool=`: the 2 x 2 average pooling behind a block comes back with the block's result, stored by the block's last launch where it can)
        _, t = self.conv32(x, pool=self.pool, pool_only=True)     # (nothing else reads this block's result)
        skips = []
        for w, _ in self.ENCODER[1:]:
            skip, t = getattr(self, "conv%d" % w)(t, pool=self.pool)
            skips.append(skip)
        t = self.conv512x512(t)
        for w in self.DECODER:
            t = getattr(self, "upsamp%d" % w)(t, residual=skips.pop())     # `t += skip` of the reference (:93-102): in the conv launch's
                                                                           # store when nothing is recorded, torch's add otherwise
            if w > 64:
                t = getattr(self, "upconv%d" % (w // 2))(t)

        outs = []
        # at inference on grayscale planes the heads' last convolutions store the row-segment layout the fused apply streams best
        blocked = (not torch.is_grad_enabled()) and gray is not None and interp_apply_gray_blocked_supported(x.shape[0], *i1.shape[2:])
        for g in (1, 2):    # heads of output channel g: frame 1 horizontal / vertical, frame 2 horizontal / vertical
            h1, v1, h2, v2 = (getattr(self, "upconv51_%d%d" % (g, k))(t, out_blocked=blocked) for k in (1, 2, 3, 4))
            if not torch.is_grad_enabled():
                # inference: pad + 2 local convolutions + add + channel mean of one output channel is one launch
                ks = (v1, h1, v2, h2)
                if gray is not None and any(k.dim() == 5 for k in ks):
                    outs.append(interp_apply_gray_blocked(gray[0], gray[1], *(k if k.dim() == 5 else coef_to_blocked(k) for k in ks)))
                elif gray is not None and interp_apply_gray_supported(*v1.shape[:1], *v1.shape[2:]):
                    outs.append(interp_apply_gray(gray[0], gray[1], v1, h1, v2, h2))
                else:
                    outs.append(interp_apply(i1, i2, v1, h1, v2, h2))
            else:           # reference :120-126
                if g == 1:
                    padded_i2, padded_i1 = self.pad(i2).contiguous(), self.pad(i1).contiguous()
                y = self.separable_conv(padded_i2, v2, h2) + self.separable_conv(padded_i1, v1, h1)
                outs.append(torch.mean(y, dim=1, keepdim=True))
        return torch.cat(outs, 1)

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

    def forward(self, x, out=None, pool=None):
        """pool: the 2 x 2 pooling module the caller applies to the result -- (result, pooled result) comes back (hipnn run_fused(pool=))."""
        if pool is not None:
            return self.double_conv(x, out=out, pool=pool)
        return self.double_conv(x, out=out) if out is not None else self.double_conv(x)


class Down(nn.Module):
    """maxpool then double conv   (reference :192-203)"""

    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.maxpool_conv = nn.Sequential(nn.MaxPool2d(2), DoubleConv(in_channels, out_channels))

    def forward(self, x, out=None, pooled=None, pool=None):
        """(nn.Sequential(MaxPool2d, DoubleConv) as in the reference, networks.py:197-200; the pooling goes through the native kernels.)
        pooled: this block's pooled input when the caller already has it (the previous block's last launch stored it: `pool=` there);
        pool: the NEXT block's pooling module -- (result, pooled result) comes back."""
        if pooled is None:
            pooled = HF.pool_module(self.maxpool_conv[0], x)
        return self.maxpool_conv[1](pooled, out=out, pool=pool)


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

    def forward(self, x1, x2, cat=None):
        if isinstance(self.up, nn.Upsample):
            # up-sample, pad to the skip's size (a no-op on even sizes: left out, torch would return a clone), cat([skip, up]) -- reference :226-231
            return self.conv(HF.skip_cat_upsample2x(self.up, x2, x1, cat=cat))
        x1 = self.up(x1)
        diffY = x2.size()[2] - x1.size()[2]
        diffX = x2.size()[3] - x1.size()[3]
        if diffY or diffX:
            x1 = F.pad(x1, [diffX // 2, diffX - diffX // 2, diffY // 2, diffY - diffY // 2])
        return self.conv(torch.cat([x2, x1], dim=1))   # skip first (reference :231)


class OutConv(nn.Module):
    def __init__(self, in_channels, out_channels):
        super(OutConv, self).__init__()
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size=1)

    def forward(self, x):
        return run_fused([self.conv], x)


class UNet(nn.Module):
    """64-128-256-512-(1024 / factor) encoder, mirrored decoder with [skip, up] concatenation (reference :241-272);
    levels built from the width table, attribute names inc / down1..4 / up1..4 / outc as in the reference."""
    WIDTHS = (64, 128, 256, 512, 1024)

    def __init__(self, n_channels, n_classes, bilinear=True):
        super().__init__()
        self.n_channels, self.n_classes, self.bilinear = n_channels, n_classes, bilinear
        factor = 2 if bilinear else 1
        w = self.WIDTHS
        self.inc = DoubleConv(n_channels, w[0])
        for k in range(1, 5):
            setattr(self, "down%d" % k, Down(w[k - 1], w[k] // (factor if k == 4 else 1)))
        for k in range(1, 5):                                   # up_k: w[5-k] channels in (skip + up), w[4-k] out
            setattr(self, "up%d" % k, Up(w[5 - k], w[4 - k] // (factor if k < 4 else 1), bilinear))
        self.outc = OutConv(w[0], n_classes)

    def forward(self, x):
        if self._skips_in_place(x):
            return self._forward_skips_in_place(x)
        feats = [self.inc(x)]
        for k in range(1, 5):
            feats.append(getattr(self, "down%d" % k)(feats[-1]))
        x = feats.pop()
        for k in range(1, 5):
            x = getattr(self, "up%d" % k)(x, feats.pop())
        return self.outc(x)

    def _skips_in_place(self, x):
        """One image, nothing recorded, bilinear decoder, sizes that halve four times: each encoder output can be stored where the
        decoder concatenates it (for one image the skip's channel block of [1, skip + up, H, W] is one contiguous run)."""
        return (self.bilinear and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[0] == 1
                and not torch.is_grad_enabled() and x.shape[2] % 16 == 0 and x.shape[3] % 16 == 0 and not self.training)

    def _forward_skips_in_place(self, x):
        # same launches and values as forward(); the four torch.cat([skip, up]) become: the skip stored in place by the encoder's
        # last launch, the up-sampled half by the up-sampling launch (hipnn.functional.skip_cat_upsample2x)
        w = self.WIDTHS
        H, W = x.shape[2], x.shape[3]
        cats, feats = [], []
        cur, pooled = x, None
        for k in range(4):                                      # encoder level k: w[k] channels at H >> k; its Up reads w[k] + w[k] channels
            cat = x.new_empty((1, 2 * w[k], H >> k, W >> k))
            skip = cat[:, :w[k]]
            nxt = getattr(self, "down%d" % (k + 1)).maxpool_conv[0]          # the next level's MaxPool2d: stored by this level's last launch
            cur, pooled = self.inc(cur, out=skip, pool=nxt) if k == 0 else getattr(self, "down%d" % k)(cur, out=skip, pooled=pooled, pool=nxt)
            cats.append(cat); feats.append(cur)
        cur = self.down4(cur, pooled=pooled)
        for k in range(1, 5):
            cur = getattr(self, "up%d" % k)(cur, feats.pop(), cat=cats.pop())
        return self.outc(cur)


class FusionNet(UNet):
    """Same network; the two inputs are added first (reference :275-306)."""

    def forward(self, x_in1, x_in2):
        return super().forward(torch.add(x_in1, x_in2))
