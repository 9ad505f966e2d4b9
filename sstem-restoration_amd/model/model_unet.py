"""SFF fusion ``UNet`` on MI355X -- same names and dataflow as the reference
``sff_scripts_fusion/model/model_unet.py:6-105`` (state_dict keys ``conv_encode1.{0,1,3,4}``,
``bottleneck.{0,1,3,4,6,7}``, ``conv_decode3.*``, ``final_layer.*``); Conv/ConvTranspose + BatchNorm +
ReLU runs execute as fused native launches (``hipnn.FusedSequential``)."""
import torch
import torch.nn as nn

from hipnn import FusedSequential
import hipnn.functional as HF


def _cbr(cin, cout):
    return [nn.Conv2d(cin, cout, kernel_size=3, padding=1), nn.BatchNorm2d(cout), nn.ReLU()]


def _up(cin, cout):
    return [nn.ConvTranspose2d(cin, cout, kernel_size=3, stride=2, padding=1, output_padding=1),
            nn.BatchNorm2d(cout), nn.ReLU()]


class UNet(nn.Module):
    def contracting_block(self, in_channels, out_channels, kernel_size=3):
        return FusedSequential(*_cbr(in_channels, out_channels), *_cbr(out_channels, out_channels))

    def expansive_block(self, in_channels, mid_channel, out_channels, kernel_size=3):
        return FusedSequential(*_cbr(in_channels, mid_channel), *_cbr(mid_channel, mid_channel),
                               *_up(mid_channel, out_channels))

    def final_block(self, in_channels, mid_channel, out_channels, kernel_size=3):
        # ends in Conv(mid->out)+BN+ReLU even for a 1-channel output (reference :42-49)
        return FusedSequential(*_cbr(in_channels, mid_channel), *_cbr(mid_channel, out_channels))

    def __init__(self, in_channel=6, out_channel=2):
        super(UNet, self).__init__()
        self.conv_encode1 = self.contracting_block(in_channels=in_channel, out_channels=32)
        self.conv_maxpool1 = nn.MaxPool2d(kernel_size=2)
        self.conv_encode2 = self.contracting_block(32, 64)
        self.conv_maxpool2 = nn.MaxPool2d(kernel_size=2)
        self.conv_encode3 = self.contracting_block(64, 128)
        self.conv_maxpool3 = nn.MaxPool2d(kernel_size=2)
        mid_channel = 128
        self.bottleneck = FusedSequential(*_cbr(mid_channel, mid_channel * 2), *_cbr(mid_channel * 2, mid_channel),
                                          *_up(mid_channel, mid_channel))
        self.conv_decode3 = self.expansive_block(256, 128, 64)
        self.conv_decode2 = self.expansive_block(128, 64, 32)
        self.final_layer = self.final_block(64, 32, out_channel)

    def crop_and_concat(self, upsampled, bypass, crop=False):
        if crop:
            c = (bypass.size()[2] - upsampled.size()[2]) // 2
            bypass = nn.functional.pad(bypass, (-c, -c, -c, -c))
        return HF.tag_concat_amax(torch.cat((upsampled, bypass), 1), upsampled, bypass)   # up-sampled first (reference :86)

    def forward(self, x):
        if self._cat_in_place(x):
            return self._forward_cat_in_place(x)
        encode_block1 = self.conv_encode1(x)
        encode_block2 = self.conv_encode2(HF.pool_module(self.conv_maxpool1, encode_block1))
        encode_block3 = self.conv_encode3(HF.pool_module(self.conv_maxpool2, encode_block2))
        bottleneck1 = self.bottleneck(HF.pool_module(self.conv_maxpool3, encode_block3))
        cat_layer2 = self.conv_decode3(self.crop_and_concat(bottleneck1, encode_block3))
        cat_layer1 = self.conv_decode2(self.crop_and_concat(cat_layer2, encode_block2))
        return self.final_layer(self.crop_and_concat(cat_layer1, encode_block1))

    @staticmethod
    def _cat_in_place(x):
        # inference on the GPU, sizes the three poolings divide: the producers store straight into the concatenated tensors
        return (not torch.is_grad_enabled()) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 \
            and x.shape[2] % 8 == 0 and x.shape[3] % 8 == 0

    def _forward_cat_in_place(self, x):
        """The same dataflow with `torch.cat((upsampled, bypass), 1)` (reference :86) done by the PRODUCERS: the tensor a decoder block
        concatenates is allocated first, the encoder block's last convolution and the ConvTranspose in front of the decoder block store
        their halves into it (hipnn.fused.run_fused(out=...); a copy where a launch cannot), and the concatenation, a read and a write
        of both tensors per level, disappears.  Same launches on the same values otherwise: same bits as the plain path."""
        N, _, H, W = x.shape
        cat1 = x.new_empty((N, 64, H, W))                   # [up 32 | encode_block1 32]
        cat2 = x.new_empty((N, 128, H // 2, W // 2))        # [up 64 | encode_block2 64]
        cat3 = x.new_empty((N, 256, H // 4, W // 4))        # [up 128 | encode_block3 128]
        e1, p1 = self.conv_encode1(x, out=cat1[:, 32:], pool=self.conv_maxpool1)         # (the pooling comes back with the block's result)
        e2, p2 = self.conv_encode2(p1, out=cat2[:, 64:], pool=self.conv_maxpool2)
        e3, p3 = self.conv_encode3(p2, out=cat3[:, 128:], pool=self.conv_maxpool3)
        u3 = self.bottleneck(p3, out=cat3[:, :128])
        HF.tag_concat_amax(cat3, u3, e3)
        u2 = self.conv_decode3(cat3, out=cat2[:, :64])
        HF.tag_concat_amax(cat2, u2, e2)
        u1 = self.conv_decode2(cat2, out=cat1[:, :32])
        HF.tag_concat_amax(cat1, u1, e1)
        return self.final_layer(cat1)
