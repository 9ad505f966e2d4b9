"""``FusedSequential``: same children, same indices, same state_dict keys as the nn.Sequential the
reference builds -- only ``forward`` differs: every run

    Conv2d(k odd, s=1, "same") | ConvTranspose2d(k3,s2,p1,op1)
        [BatchNorm2d]            (folded into the launch when the BN is in eval mode)
        [ReLU | LeakyReLU]       (fused unless a train-mode BN sits in between)

becomes one native launch (hipnn.functional).  A train-mode BatchNorm needs the whole batch before it can normalise
("Hard parts"): there the conv launch is followed by the native batch-statistics + normalise +
activation passes (include/sstem_norm.h).  A bilinear x2 Upsample(align_corners=True) on planes up to 256x256
is one native launch (native gather backward when recording); any other child (pooling, ...) runs as it is.
"""
import torch
import torch.nn as nn

from . import functional as F_


def _is_same_conv(m):
    return (isinstance(m, nn.Conv2d) and m.stride == (1, 1) and m.dilation == (1, 1) and m.groups == 1
            and m.kernel_size[0] == m.kernel_size[1] and m.kernel_size[0] % 2 == 1
            and m.padding == (m.kernel_size[0] // 2, m.kernel_size[1] // 2) and m.padding_mode == "zeros")


def _is_up_convT(m):
    return (isinstance(m, nn.ConvTranspose2d) and m.kernel_size == (3, 3) and m.stride == (2, 2)
            and m.padding == (1, 1) and m.output_padding == (1, 1) and m.groups == 1 and m.dilation == (1, 1))


def _act_of(m):
    if isinstance(m, nn.ReLU):
        return F_.ACT_RELU, 0.0
    if isinstance(m, nn.LeakyReLU):
        return F_.ACT_LEAKY, float(m.negative_slope)
    return None


def _bn_affine(bn):
    """Eval-mode BatchNorm as y = x*scale + shift.  Cached on the module: in eval mode the four tensors only change when
    someone loads or edits them, which bumps their version counters (recomputing the fold on every forward was five tiny
    launches per BatchNorm layer: 0.9 ms per fusion step for the frozen 49-BN flow network)."""
    key = (bn.running_mean._version, bn.running_var._version, bn.running_mean.data_ptr(), bn.running_var.data_ptr(),
           (bn.weight._version, bn.bias._version, bn.weight.data_ptr(), bn.bias.data_ptr()) if bn.affine else None, bn.eps)
    if F_._touch_log is not None:
        F_._touch_log.append(tuple(t for t in (bn.running_mean, bn.running_var, bn.weight, bn.bias) if t is not None))
    cached = getattr(bn, "_sstem_fold", None)
    if cached is not None and cached[0] == key:
        return cached[1], cached[2]
    with torch.no_grad():
        inv = torch.rsqrt(bn.running_var + bn.eps)
        scale = inv * bn.weight if bn.affine else inv
        shift = -bn.running_mean * scale
        if bn.affine:
            shift = shift + bn.bias
        scale, shift = scale.contiguous(), shift.contiguous()
    bn._sstem_fold = (key, scale, shift)
    return scale, shift


def run_fused(children, x, residual=None, res_scale=1.0, out=None, out_blocked=False, pool=None, pool_only=False):
    """Run a list of modules with Conv(+BN eval)(+act) groups fused into single launches.
    out (optional, only when nothing is recorded for a backward): an fp32 tensor of the result's shape, contiguous or a channel block of
    a larger contiguous NCHW tensor; the result is stored there and `out` is returned -- by the last launch itself when that is a fused
    fp32 convolution that can (a contiguous `out`: any such launch; a channel block: a split-kernel launch through the scaled entry, or
    the sub-pixel form of a ConvTranspose -- hipnn.functional.strided_store_ok; the U-Nets place their encoder outputs and up-sampled
    tensors inside the tensors their decoders concatenate), by a copy otherwise.
    out_blocked: a REQUEST to store the result in the row-segment layout [N, H, ceil(W/64), C, 64] (the blocked coefficients the fused
    sepconv apply reads): granted when the last group is a 3x3 convolution launch that can (hipnn.functional.blocked_store_ok) -- the
    result then has five dimensions --, otherwise the result is the usual NCHW tensor.
    residual (optional): the result is ``(children(x) + residual) * res_scale`` -- the additive skips of the reference's blocks
    (model_fusionnet.py:57-61, :129-138).  When the LAST group is a fused fp32 convolution launch and nothing is being recorded
    for a backward, the add and the scale happen in that launch's store; otherwise they are the two torch operations the
    reference runs."""
    i, n = 0, len(children)
    pending_residual = residual is not None
    stored = False
    pooled = None           # pool (optional): the 2 x 2 pooling module the caller applies to the result -- (result, pooled result) is returned;
                            # the last launch stores the pooled copy itself when it can (hipnn.functional.pooled_store_ok)
                            # pool_only: the caller reads nothing but the pooled result -- (None, pooled result) when the last launch can
                            # store the pooled copy alone, (result, pooled result) otherwise
    while i < n:
        m = children[i]
        conv_like = _is_same_conv(m) or _is_up_convT(m)
        if not conv_like:
            if x.dtype == torch.bfloat16:
                x = x.float()                                   # (not produced today: bf16 is only handed to a following 3x3 conv)
            if F_.is_bilinear2x(m):
                x = F_.upsample_bilinear2x_module(m, x)      # native forward on the planes where it wins (native backward too)
            elif isinstance(m, (nn.MaxPool2d, nn.AvgPool2d)):
                x = F_.pool_module(m, x)                     # the 2 x 2 pooling of the reference's networks: native streaming kernels
            else:
                x = m(x)
            i += 1
            continue
        # bf16 id, recording: a run of Conv3x3 [+ activation] groups without BatchNorm becomes one autograd function whose inner
        # tensors are bf16 (hipnn.functional._ConvChain)
        if isinstance(m, nn.Conv2d) and m.kernel_size == (3, 3) and torch.is_grad_enabled() and F_.get_algorithm() == F_.ALGO_MFMA_BF16:
            convs, spec, k = [], [], i
            while k < n and _is_same_conv(children[k]) and children[k].kernel_size == (3, 3):
                nk = k + 1
                if nk < n and isinstance(children[nk], nn.BatchNorm2d):
                    break
                a = _act_of(children[nk]) if nk < n else None
                convs.append(children[k]); spec.append(a if a is not None else (F_.ACT_NONE, 0.0))
                k = nk + (1 if a is not None else 0)
            if len(convs) >= 2 and F_.conv_chain_ok(x, convs):
                x = F_.conv_chain(x, convs, spec)
                i = k
                continue
        fn = F_.conv2d_fused if isinstance(m, nn.Conv2d) else F_.conv_transpose3x3s2_fused
        j = i + 1
        scale = shift = None
        bn = None
        if j < n and isinstance(children[j], nn.BatchNorm2d):
            bn = children[j]
            j += 1
        act = None
        if j < n:
            act = _act_of(children[j])
            if act is not None:
                j += 1
        if bn is not None and (bn.training or not bn.track_running_stats):
            # batch statistics needed: conv launch, then batch statistics + normalisation + activation as two native
            # streaming passes (torch's train-mode BatchNorm + ReLU is 4-5x off the streaming bound on the full-resolution
            # layers: tools/bench_bn.py)
            if x.dtype == torch.bfloat16:
                x = x.float()
            native_bn = x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and fn_cout(m) <= 65535
            # the conv launch writes the batch-statistics partials of its own output while it stores it (one pass over the tensor less)
            parts = F_.bn_partials_for(x, m) if native_bn else None
            x = fn(x, m.weight, m.bias, owner=m, bn_part=parts)
            if native_bn:
                a, slope = act if act is not None else (F_.ACT_NONE, 0.0)
                x = F_.batchnorm_train_act(bn, x, a, slope, partials=parts)
            else:
                x = bn(x)
                if act is not None:
                    x = children[j - 1](x)
        elif bn is not None and torch.is_grad_enabled() and (x.requires_grad or m.weight.requires_grad or
                                                              (bn.affine and (bn.weight.requires_grad or bn.bias.requires_grad))):
            # an eval-mode BatchNorm with a gradient being recorded (fine-tuning with frozen statistics; torch -- and so the reference's
            # modules -- allow it): the folded launch has no backward, so the convolution runs unfolded and the BatchNorm and the
            # activation are the modules themselves (torch's eval-mode batch_norm is differentiable in x, weight and bias)
            if x.dtype == torch.bfloat16:
                x = x.float()
            x = bn(fn(x, m.weight, m.bias, owner=m))
            if act is not None:
                x = children[j - 1](x)
        else:
            if bn is not None:
                scale, shift = _bn_affine(bn)
            a, slope = act if act is not None else (F_.ACT_NONE, 0.0)
            # under the bf16 id, with no backward possible, the convolutions inside this Sequential hand each other bf16 tensors
            # (numerically free: the consumer rounds the same values anyway; half the traffic)
            nxt = children[j] if j < n else None
            to_bf16 = nxt is not None and _is_same_conv(nxt) and nxt.kernel_size == (3, 3)
            if isinstance(m, nn.Conv2d) and F_.bf16io_ok(x, m, False) and (x.dtype == torch.bfloat16 or (to_bf16 and F_.bf16io_ok(x, m, True))):
                ob = to_bf16 and F_.bf16io_ok(x, m, True)
                into = out if (out is not None and j == n and not pending_residual and not ob) else None
                x = F_.conv3x3_bf16io(x, m.weight, m.bias, scale, shift, a, slope, out_bf16=ob, owner=m, out=into)
                stored = stored or into is not None
            else:
                if x.dtype == torch.bfloat16:
                    x = x.float()
                if pending_residual and j == n and F_.residual_fusable(x, m, residual):
                    x = fn(x, m.weight, m.bias, scale, shift, a, slope, owner=m, residual=residual, res_scale=res_scale)
                    pending_residual = False
                elif pool is not None and j == n and not pending_residual and (out is None or (F_.can_store_into(x, m.weight, m.bias)
                                                                                      and F_.strided_store_ok(x, m, out))) \
                        and F_.pooled_store_ok(x, m, pool):
                    kind = F_.pooled_store_ok(x, m, pool)
                    pooled = x.new_empty((x.shape[0], fn_cout(m), x.shape[2] // 2, x.shape[3] // 2))
                    if pool_only and out is None and isinstance(m, nn.Conv2d):      # nobody reads the full-resolution result: it is not stored
                        fn(x, m.weight, m.bias, scale, shift, a, slope, owner=m, pool_out=pooled, pool_kind=kind, pool_only=True)
                        x = None
                    else:
                        x = fn(x, m.weight, m.bias, scale, shift, a, slope, owner=m, out=out, pool_out=pooled, pool_kind=kind)
                    stored = out is not None
                elif out is not None and j == n and not pending_residual and isinstance(m, nn.Conv2d) and F_.can_store_into(x, m.weight, m.bias) \
                        and F_.strided_store_ok(x, m, out):
                    x = fn(x, m.weight, m.bias, scale, shift, a, slope, owner=m, out=out)
                    stored = True
                elif out is not None and j == n and not pending_residual and isinstance(m, nn.ConvTranspose2d) and F_.strided_store_ok(x, m, out):          # the sub-pixel form stores into the consumer's concatenated tensor
                    x = fn(x, m.weight, m.bias, scale, shift, a, slope, owner=m, out=out)
                    stored = True
                elif out_blocked and out is None and j == n and not pending_residual and isinstance(m, nn.Conv2d) and F_.blocked_store_ok(x, m):
                    x = fn(x, m.weight, m.bias, scale, shift, a, slope, owner=m, out_blocked=True)
                else:
                    x = fn(x, m.weight, m.bias, scale, shift, a, slope, owner=m)
        i = j
    if pending_residual:
        if x.dtype == torch.bfloat16:
            x = x.float()
        parts = (x, residual)
        x = x + residual
        if res_scale != 1.0:
            x = x * res_scale
        F_.tag_sum_amax(x, parts[0], parts[1], res_scale)    # (the next layer's fp16 launch scales by it instead of measuring x)
    if out is not None and not stored:
        out.copy_(x)
        x = out
    if pool is not None:
        if pooled is None:
            pooled = F_.pool_module(pool, x)
        return x, pooled
    return x


def fn_cout(m):
    return m.weight.shape[1] if isinstance(m, nn.ConvTranspose2d) else m.weight.shape[0]


def invalidate_caches(module):
    """Drop what hipnn keeps on the modules under `module` (packed conv weights of frozen / inference calls, folded eval-mode
    BatchNorm).  The caches follow torch's version counters, so they need this only when parameters or buffers were changed where
    torch cannot see it: a replayed HIP graph that contains the optimizer step, or writes through raw pointers by foreign code."""
    for m in module.modules():
        m.__dict__.pop("_sstem_packs", None)
        m.__dict__.pop("_sstem_fold", None)


class FusedSequential(nn.Sequential):
    def forward(self, x, residual=None, res_scale=1.0, out=None, out_blocked=False, pool=None, pool_only=False):
        return run_fused(list(self), x, residual, res_scale, out, out_blocked, pool, pool_only)
