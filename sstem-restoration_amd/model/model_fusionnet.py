"""SFF ``FusionNet`` (the frozen flow predictor of the fusion stage) on MI355X -- same names and
dataflow as the reference ``sff_scripts_fusion/model/model_fusionnet.py:12-145`` (state_dict keys
``down_1.conv_1.{0,1}``, ``down_1.conv_2.{0.0,0.1,1.0,1.1,2,3}``, ``deconv_1.{0,1}``, ``out`` ...).
Conv3x3+BN+LeakyReLU(0.2) encoder blocks, ConvTranspose+BN+ReLU decoder blocks and the residual
3-conv blocks run as fused native launches; in eval mode (how the fusion stage uses it,
main_fusion.py:189,227-228) BatchNorm folds into the launch."""
import torch.nn as nn

from hipnn import FusedSequential
import hipnn.functional as HF
from hipnn.fused import run_fused


def conv_block(in_dim, out_dim, act_fn):
    return FusedSequential(nn.Conv2d(in_dim, out_dim, kernel_size=3, stride=1, padding=1),
                           nn.BatchNorm2d(out_dim), act_fn)


def conv_trans_block(in_dim, out_dim, act_fn):
    return FusedSequential(nn.ConvTranspose2d(in_dim, out_dim, kernel_size=3, stride=2, padding=1, output_padding=1),
                           nn.BatchNorm2d(out_dim), act_fn)


def maxpool():
    return nn.MaxPool2d(kernel_size=2, stride=2, padding=0)


class _Block3(nn.Sequential):
    """conv_block, conv_block, Conv2d, BatchNorm2d (reference conv_block_3, :36-43): children 0 and 1 are
    fused sequentials themselves; 2 and 3 fuse here (BN folded in eval mode, no activation)."""

    def forward(self, x, residual=None):
        x = self[0](x)
        x = self[1](x)
        return run_fused([self[2], self[3]], x, residual)


def conv_block_3(in_dim, out_dim, act_fn):
    return _Block3(conv_block(in_dim, out_dim, act_fn), conv_block(out_dim, out_dim, act_fn),
                   nn.Conv2d(out_dim, out_dim, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(out_dim))


class Conv_residual_conv(nn.Module):
    """conv_1 -> (conv_2: three convs, last one without activation) -> add -> conv_3   (reference :45-62)."""

    def __init__(self, in_dim, out_dim, act_fn):
        super().__init__()
        self.in_dim, self.out_dim = in_dim, out_dim
        self.conv_1 = conv_block(in_dim, out_dim, act_fn)
        self.conv_2 = conv_block_3(out_dim, out_dim, act_fn)
        self.conv_3 = conv_block(out_dim, out_dim, act_fn)

    def forward(self, x, pool=None):
        """pool: the 2 x 2 pooling module the caller applies to the block's output -- (output, pooled output) comes back, the pooled copy
        stored by conv_3's launch itself where it can (hipnn.fused.run_fused(pool=...))."""
        head = self.conv_1(x)
        return self.conv_3(self.conv_2(head, residual=head), pool=pool)        # conv_1 + conv_2 (reference :57-61), added in conv_2's last store


class FusionNet(nn.Module):
    """Four encoder levels (LeakyReLU 0.2), a bridge, four decoder levels (ReLU); decoder level k averages its transposed-conv
    output with the encoder output of the same resolution.  The levels are built from a width table; the attribute names
    (down_k, pool_k, bridge, deconv_k, up_k, out) and their registration order are the reference's, so its checkpoints load
    with strict=True."""
    LEVELS = 4

    def __init__(self, input_nc=6, output_nc=2, ngf=32):
        super().__init__()
        self.in_dim, self.out_dim, self.final_out_dim = input_nc, ngf, output_nc
        leaky, relu = nn.LeakyReLU(0.2, inplace=True), nn.ReLU()
        widths = [ngf << k for k in range(self.LEVELS + 1)]               # ngf, 2 ngf, ... 16 ngf
        c = input_nc
        for k in range(self.LEVELS):
            setattr(self, "down_%d" % (k + 1), Conv_residual_conv(c, widths[k], leaky))
            setattr(self, "pool_%d" % (k + 1), maxpool())
            c = widths[k]
        self.bridge = Conv_residual_conv(c, widths[-1], leaky)
        for k in range(self.LEVELS):                                      # 16 ngf -> 8 ngf -> ... -> ngf
            w = widths[self.LEVELS - 1 - k]
            setattr(self, "deconv_%d" % (k + 1), conv_trans_block(2 * w, w, relu))
            setattr(self, "up_%d" % (k + 1), Conv_residual_conv(w, w, relu))
        self.out = nn.Conv2d(ngf, output_nc, kernel_size=3, stride=1, padding=1)
        self.apply(self._reference_init)

    @staticmethod
    def _reference_init(m):
        # reference :107-113: conv weights N(0, 0.02) with zero bias, BatchNorm scale N(1, 0.02) with zero shift
        if isinstance(m, nn.Conv2d):
            m.weight.data.normal_(0.0, 0.02); m.bias.data.fill_(0)
        elif isinstance(m, nn.BatchNorm2d):
            m.weight.data.normal_(1.0, 0.02); m.bias.data.fill_(0)

    def forward(self, x):
        skips = []
        for k in range(1, self.LEVELS + 1):
            skip, x = getattr(self, "down_%d" % k)(x, pool=getattr(self, "pool_%d" % k))
            skips.append(skip)
        x = self.bridge(x)
        for k in range(1, self.LEVELS + 1):
            # (deconv + down) / 2 (reference :129-138): in the store of the transposed convolution's launch when nothing is recorded
            x = getattr(self, "up_%d" % k)(getattr(self, "deconv_%d" % k)(x, residual=skips[-k], res_scale=0.5))
        return run_fused([self.out], x)
