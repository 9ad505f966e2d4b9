"""SFF interpolation network ``IFNet`` on MI355X.

Same class name, constructor, sub-module names (hence ``state_dict`` keys: ``conv32.{0,2,4}``,
``upsamp512.1``, ``upconv51_1.{0,2,4,7}``, ``srconv1..4`` ...) and forward dataflow as the reference
``sff_scripts_interp/model/model_interp.py:9-148``.  The Conv3x3+ReLU runs are fused launches of the
gfx950 convolution kernel (``hipnn.FusedSequential``) and the local convolutions are the native
sepconv op; pooling / bilinear up-sampling / replication padding stay torch device ops.
"""
import torch
import torch.nn as nn
import torch.nn.init as init

from hipnn import FusedSequential
from hipnn.fused import run_fused
import hipnn.functional as HF
from libs.sepconv.SeparableConvolution import SeparableConvolution
from libs.sepconv.fused import (coef_to_blocked, interp_apply, interp_apply_gray, interp_apply_gray_blocked,
                                interp_apply_gray_blocked_supported, interp_apply_gray_supported, interp_apply_gray_u8)


def _conv3(cin, cout):
    return nn.Conv2d(cin, cout, (3, 3), (1, 1), 1)


class IFNet(nn.Module):
    def __init__(self, kernel_size=51):
        super(IFNet, self).__init__()
        taps = kernel_size

        self.pool = nn.AvgPool2d(kernel_size=(2, 2), stride=(2, 2))
        self.upsamp = nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)
        self.relu = nn.ReLU(inplace=False)

        # encoder / decoder trunk (reference :22-34); note the in->in->in->out widths of _conv_module
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
        # four kernel-prediction heads (reference :35-38)
        self.upconv51_1 = self._kernel_module(64, taps)
        self.upconv51_2 = self._kernel_module(64, taps)
        self.upconv51_3 = self._kernel_module(64, taps)
        self.upconv51_4 = self._kernel_module(64, taps)

        # registered but unused by forward (reference :39-44,53): kept for checkpoint compatibility
        upscale_factor = 2
        self.srconv1 = nn.Conv2d(1, 64, (5, 5), (1, 1), (2, 2))
        self.srconv2 = nn.Conv2d(64, 64, (3, 3), (1, 1), (1, 1))
        self.srconv3 = nn.Conv2d(64, 32, (3, 3), (1, 1), (1, 1))
        self.srconv4 = nn.Conv2d(32, upscale_factor ** 2, (3, 3), (1, 1), (1, 1))
        self.pixel_shuffle = nn.PixelShuffle(upscale_factor)

        self.pad = nn.ReplicationPad2d(taps // 2)
        self.separable_conv = SeparableConvolution.apply

        self.apply(self._weight_init)
        self.sigmoid = nn.Sigmoid()

    def forward(self, x):
        return self._interpolate(x, None)

    def interpolate_gray(self, frame1, frame2):
        """``forward`` for callers that hold the two grayscale planes [B,1,H,W] and would build the network input by
        replicating each x3 (reference inference_singleImage.py:55-66).  Same result, bit for bit; at inference the local
        convolutions then run as one launch on the planes themselves (no channel comparison on the device)."""
        B, _, H, W = frame1.shape
        x = torch.cat((frame1.expand(B, 3, H, W), frame2.expand(B, 3, H, W)), 1)
        return self._interpolate(x, (frame1, frame2))

    def interpolate_gray_u8(self, frames_u8):
        """Inference from the uint8 frames themselves (reference inference_singleImage.py:55-66,76): ``frames_u8``
        [B,2,H,W] uint8 = the two grayscale sections as read from their PNGs.  The first convolution reads the bytes (value / 255 in
        float32, the x3 replication implied: ``sstem_conv3x3_first_layer_u8``) and leaves the two float32 planes behind for the local
        convolutions; the fused apply stores ``(pred * 255).astype(uint8)`` next to ``pred``.  Returns (pred float32 [B,1,H,W],
        image uint8 [B,H,W]) -- bit for bit ``interpolate_gray`` on ``frames_u8 / 255`` and numpy's truncation of its result."""
        if torch.is_grad_enabled():
            raise RuntimeError("interpolate_gray_u8 is an inference path: call it under torch.no_grad()")
        if not HF.first_layer_u8_ok(frames_u8, self.conv32[0]):
            raise NotImplementedError("interpolate_gray_u8: uint8 GPU frames [B,2,H,W] with W % 4 == 0 (use interpolate_gray on frames / 255)")
        first, planes = HF.first_layer_u8(frames_u8, self.conv32[0], HF.ACT_RELU)
        g1, g2 = planes[0], planes[1]                              # [B,1,H,W] each, contiguous (the launch stores them frame-major)
        _, x = run_fused(list(self.conv32)[2:], first, pool=self.pool, pool_only=True)
        return self._interpolate(None, (g1, g2), pooled32=x, want_u8=True)

    def _interpolate(self, x, gray, pooled32=None, want_u8=False):
        if x is not None:
            i1 = x[:, :3]
            i2 = x[:, 3:6]

        # contraction (reference :60-70)
        # (`pool=`: the 2 x 2 average pooling behind a block comes back with the block's result -- stored by the block's last launch
        #  itself where that launch can, by the pooling kernel otherwise: same values, same bits)
        if pooled32 is not None:
            x = pooled32                                          # (the first block ran from the uint8 frames: interpolate_gray_u8)
        else:
            _, x = self.conv32(x, pool=self.pool, pool_only=True)     # (nothing else reads this block's result)
        x64, x = self.conv64(x, pool=self.pool)
        x128, x = self.conv128(x, pool=self.pool)
        x256, x = self.conv256(x, pool=self.pool)
        x512, x = self.conv512(x, pool=self.pool)
        x = self.conv512x512(x)

        # expansion with additive skips (reference :73-83: `x = self.upsamp512(x); x += x512` ...).  FusedSequential adds the skip in
        # the store of the module's convolution launch when nothing is recorded for a backward, with torch's add otherwise
        x = self.upsamp512(x, residual=x512)
        x = self.upconv256(x)
        x = self.upsamp256(x, residual=x256)
        x = self.upconv128(x)
        x = self.upsamp128(x, residual=x128)
        x = self.upconv64(x)
        x = self.upsamp64(x, residual=x64)

        # per-pixel 51-tap kernels (reference :86-89)
        # at inference on grayscale planes the heads' last convolutions store the row-segment layout the fused apply streams best
        # (include/sstem_sepconv.h, "blocked coefficients": same values, same bits out of the apply)
        hw = gray[0].shape[2:] if gray is not None else i1.shape[2:]
        blocked = (not torch.is_grad_enabled()) and gray is not None and interp_apply_gray_blocked_supported(x.shape[0], *hw)
        k2h = self.upconv51_1(x, out_blocked=blocked)
        k2v = self.upconv51_2(x, out_blocked=blocked)
        k1h = self.upconv51_3(x, out_blocked=blocked)
        k1v = self.upconv51_4(x, out_blocked=blocked)
        if not torch.is_grad_enabled():
            # inference: pad + both local convolutions + add + channel mean in one launch
            ks = (k1v, k1h, k2v, k2h)
            if want_u8:
                if any(k.dim() == 5 for k in ks):
                    ks = tuple(k if k.dim() == 5 else coef_to_blocked(k) for k in ks)
                return interp_apply_gray_u8(gray[0], gray[1], *ks)
            if gray is not None and any(k.dim() == 5 for k in ks):
                return interp_apply_gray_blocked(gray[0], gray[1], *(k if k.dim() == 5 else coef_to_blocked(k) for k in ks))
            if gray is not None and interp_apply_gray_supported(*k1v.shape[:1], *k1v.shape[2:]):
                return interp_apply_gray(gray[0], gray[1], k1v, k1h, k2v, k2h)
            return interp_apply(i1, i2, k1v, k1h, k2v, k2h)

        padded_i2 = self.pad(i2).contiguous()
        padded_i1 = self.pad(i1).contiguous()

        # local convolutions + channel mean (reference :94-97)
        y = self.separable_conv(padded_i2, k2v, k2h) + self.separable_conv(padded_i1, k1v, k1h)
        return torch.mean(y, dim=1, keepdim=True)

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
