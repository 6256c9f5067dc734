"""Oracle (TEST INFRASTRUCTURE -- only tests/ may import this): the float16-operand training arithmetic.

The reference trains under `torch.cuda.amp.autocast()` + `GradScaler` when `--use_amp` is set (src/train.py:246-262,
src/params.py:71): its convolutions see float16 inputs and weights and accumulate in fp32, in the forward pass and in both
gradients.  The hand-written f16 training kernels (`mst_encoder_set_train_precision(enc, 1)`, include/mst.h) define the
same thing precisely: BOTH operands of every convolution-shaped product are rounded to float16 --

    forward          y  = r(conv(r(x), r(w)) + b)        (the output is stored as float16, as autocast's conv output is;
                                                          BatchNorm's batch statistics are those of the stored values)
    input gradient   dx = conv^T(r(dy), r(w))
    weight gradient  dW = corr(r(x), r(dy))

-- and everything else is exact (the rounding of y is transparent to the gradient, as a dtype cast is in autograd).  This module restates that as a `torch.autograd.Function`, to be evaluated in float64 on
the CPU or the GPU by the tests: r() rounds to float16's 11 significant bits with an UNBOUNDED exponent, because the
kernels multiply every tensor by an exact power of two before rounding (per-channel weight scale, per-band activation
scale, one loss scale per backward pass) so that neither the 65504 ceiling nor the subnormals are reached.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


from .encoder import round_f16_ideal  # noqa: E402  (11 significant bits, unbounded exponent)


class _F16OperandConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, padding):
        xr, wr = round_f16_ideal(x), round_f16_ideal(w)
        ctx.save_for_backward(xr, wr)
        ctx.padding = padding
        return round_f16_ideal(F.conv2d(xr, wr, b, padding=padding))   # the output is STORED as float16 (gradient: straight through)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr = round_f16_ideal(dy)
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, padding=ctx.padding) if ctx.needs_input_grad[0] else None
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, padding=ctx.padding)
        return dx, dw, dy.sum(dim=(0, 2, 3)), None


class F16OperandConv2d(nn.Conv2d):
    """nn.Conv2d whose forward and both gradients round their two operands to float16 precision (see module docstring)"""

    def forward(self, x):
        return _F16OperandConv.apply(x, self.weight, self.bias, self.padding)


def convert_convs(model: nn.Module) -> nn.Module:
    """switch every 7x7 convolution of the sub-band CNNs (src/model.py:107-125) to the f16-operand arithmetic, in place"""
    for c in model.audio_encoder.subnet_cnns:
        c.conv1.__class__ = F16OperandConv2d
        c.conv2.__class__ = F16OperandConv2d
    return model
