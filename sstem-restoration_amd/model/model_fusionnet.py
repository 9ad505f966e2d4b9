"""SFF ``FusionNet`` (the frozen flow predictor of the fusion stage) on MI355X -- same names and
dataflow as the reference ``sff_scripts_fusion/model/model_fusionnet.py:12-145`` (state_dict keys
``down_1.conv_1.{0,1}``, ``down_1.conv_2.{0.0,0.1,1.0,1.1,2,3}``, ``deconv_1.{0,1}``, ``out`` ...).
Conv3x3+BN+LeakyReLU(0.2) encoder blocks, ConvTranspose+BN+ReLU decoder blocks and the residual
3-conv blocks run as fused native launches; in eval mode (how the fusion stage uses it,
main_fusion.py:189,227-228) BatchNorm folds into the launch."""
import torch.nn as nn

from hipnn import FusedSequential
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

    def forward(self, x):
        x = self[0](x)
        x = self[1](x)
        return run_fused([self[2], self[3]], x)


def conv_block_3(in_dim, out_dim, act_fn):
    return _Block3(conv_block(in_dim, out_dim, act_fn), conv_block(out_dim, out_dim, act_fn),
                   nn.Conv2d(out_dim, out_dim, kernel_size=3, stride=1, padding=1), nn.BatchNorm2d(out_dim))


class Conv_residual_conv(nn.Module):
    def __init__(self, in_dim, out_dim, act_fn):
        super(Conv_residual_conv, self).__init__()
        self.in_dim = in_dim
        self.out_dim = out_dim
        self.conv_1 = conv_block(self.in_dim, self.out_dim, act_fn)
        self.conv_2 = conv_block_3(self.out_dim, self.out_dim, act_fn)
        self.conv_3 = conv_block(self.out_dim, self.out_dim, act_fn)

    def forward(self, input):
        conv_1 = self.conv_1(input)
        conv_2 = self.conv_2(conv_1)
        return self.conv_3(conv_1 + conv_2)


class FusionNet(nn.Module):
    def __init__(self, input_nc=6, output_nc=2, ngf=32):
        super(FusionNet, self).__init__()
        self.in_dim = input_nc
        self.out_dim = ngf
        self.final_out_dim = output_nc
        act_fn = nn.LeakyReLU(0.2, inplace=True)
        act_fn_2 = nn.ReLU()

        self.down_1 = Conv_residual_conv(self.in_dim, self.out_dim, act_fn)
        self.pool_1 = maxpool()
        self.down_2 = Conv_residual_conv(self.out_dim, self.out_dim * 2, act_fn)
        self.pool_2 = maxpool()
        self.down_3 = Conv_residual_conv(self.out_dim * 2, self.out_dim * 4, act_fn)
        self.pool_3 = maxpool()
        self.down_4 = Conv_residual_conv(self.out_dim * 4, self.out_dim * 8, act_fn)
        self.pool_4 = maxpool()

        self.bridge = Conv_residual_conv(self.out_dim * 8, self.out_dim * 16, act_fn)

        self.deconv_1 = conv_trans_block(self.out_dim * 16, self.out_dim * 8, act_fn_2)
        self.up_1 = Conv_residual_conv(self.out_dim * 8, self.out_dim * 8, act_fn_2)
        self.deconv_2 = conv_trans_block(self.out_dim * 8, self.out_dim * 4, act_fn_2)
        self.up_2 = Conv_residual_conv(self.out_dim * 4, self.out_dim * 4, act_fn_2)
        self.deconv_3 = conv_trans_block(self.out_dim * 4, self.out_dim * 2, act_fn_2)
        self.up_3 = Conv_residual_conv(self.out_dim * 2, self.out_dim * 2, act_fn_2)
        self.deconv_4 = conv_trans_block(self.out_dim * 2, self.out_dim, act_fn_2)
        self.up_4 = Conv_residual_conv(self.out_dim, self.out_dim, act_fn_2)

        self.out = nn.Conv2d(self.out_dim, self.final_out_dim, kernel_size=3, stride=1, padding=1)

        # reference initialisation (:107-113): N(0,0.02) conv weights, zero biases, BN N(1,0.02)/0
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                m.weight.data.normal_(0.0, 0.02)
                m.bias.data.fill_(0)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.normal_(1.0, 0.02)
                m.bias.data.fill_(0)

    def forward(self, input):
        down_1 = self.down_1(input)
        down_2 = self.down_2(self.pool_1(down_1))
        down_3 = self.down_3(self.pool_2(down_2))
        down_4 = self.down_4(self.pool_3(down_3))
        bridge = self.bridge(self.pool_4(down_4))

        up_1 = self.up_1((self.deconv_1(bridge) + down_4) / 2)
        up_2 = self.up_2((self.deconv_2(up_1) + down_3) / 2)
        up_3 = self.up_3((self.deconv_3(up_2) + down_2) / 2)
        up_4 = self.up_4((self.deconv_4(up_3) + down_1) / 2)
        return run_fused([self.out], up_4)
