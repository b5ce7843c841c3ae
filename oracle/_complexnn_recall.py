"""Stand-in for the reference's ABSENT third-party `complexnn` module.

TEST INFRASTRUCTURE, GOLDEN GENERATION ONLY (build container).

`DCCRN/DCCRN_cprs.py:6` imports `ComplexConv2d, ComplexConvTranspose2d,
NavieComplexLSTM, complex_cat, ComplexBatchNorm` from a `complexnn.py` that is
not in /root/reference, not pip-installed, and not version-pinned anywhere in
the reference.  Its origin is the upstream DCCRN repository
(huyanxin/DeepComplexCRN, `complexnn.py`).  This file RESTATES that published
module's behaviour, written from its documented semantics and the call-site
contract in DCCRN_cprs.py:

  * channel axis holds [real half ; imag half] (`complex_axis=1`);
  * ComplexConv2d = two real nn.Conv2d (`real_conv`, `imag_conv`), frequency
    padding symmetric, time padding `padding[1]` on the LEFT only (causal);
    out_real = real_conv(r) - imag_conv(i); out_imag = imag_conv(r) + real_conv(i);
  * ComplexConvTranspose2d = two real nn.ConvTranspose2d, same combination;
  * NavieComplexLSTM(input_size, hidden_size, projection_dim) = two real
    nn.LSTM(input_size//2, hidden_size//2) (`real_lstm`, `imag_lstm`),
    real = real_lstm(r) - imag_lstm(i); imag = real_lstm(i) + imag_lstm(r);
    optional per-part nn.Linear `r_trans` / `i_trans`;
  * complex_cat concatenates real halves together and imag halves together.

Because this is a restatement and not the upstream file, DCCRN parity is
PINNED ABOVE this boundary (the reference's own DCCRN class runs on top of it)
and UNPINNED AT it.  It only exists so `oracle/gen_golden.py` can import the
reference's DCCRN class; nothing on the product path uses it.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class ComplexConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 dilation=1, groups=1, causal=True, complex_axis=1):
        super().__init__()
        self.padding = padding
        self.causal = causal
        self.complex_axis = complex_axis
        self.real_conv = nn.Conv2d(in_channels // 2, out_channels // 2, kernel_size, stride,
                                   padding=[padding[0], 0], dilation=dilation, groups=groups)
        self.imag_conv = nn.Conv2d(in_channels // 2, out_channels // 2, kernel_size, stride,
                                   padding=[padding[0], 0], dilation=dilation, groups=groups)

    def forward(self, inputs):
        if self.padding[1] != 0 and self.causal:
            inputs = F.pad(inputs, [self.padding[1], 0, 0, 0])
        else:
            inputs = F.pad(inputs, [self.padding[1], self.padding[1], 0, 0])
        real, imag = torch.chunk(inputs, 2, self.complex_axis)
        real_out = self.real_conv(real) - self.imag_conv(imag)
        imag_out = self.imag_conv(real) + self.real_conv(imag)
        return torch.cat([real_out, imag_out], self.complex_axis)


class ComplexConvTranspose2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 output_padding=(0, 0), causal=False, complex_axis=1, groups=1):
        super().__init__()
        self.complex_axis = complex_axis
        self.real_conv = nn.ConvTranspose2d(in_channels // 2, out_channels // 2, kernel_size, stride,
                                            padding=padding, output_padding=output_padding, groups=groups)
        self.imag_conv = nn.ConvTranspose2d(in_channels // 2, out_channels // 2, kernel_size, stride,
                                            padding=padding, output_padding=output_padding, groups=groups)

    def forward(self, inputs):
        real, imag = torch.chunk(inputs, 2, self.complex_axis)
        real_out = self.real_conv(real) - self.imag_conv(imag)
        imag_out = self.imag_conv(real) + self.real_conv(imag)
        return torch.cat([real_out, imag_out], self.complex_axis)


class NavieComplexLSTM(nn.Module):
    def __init__(self, input_size, hidden_size, projection_dim=None, bidirectional=False, batch_first=False):
        super().__init__()
        self.real_lstm = nn.LSTM(input_size // 2, hidden_size // 2, num_layers=1,
                                 bidirectional=bidirectional, batch_first=False)
        self.imag_lstm = nn.LSTM(input_size // 2, hidden_size // 2, num_layers=1,
                                 bidirectional=bidirectional, batch_first=False)
        d = 2 if bidirectional else 1
        self.projection_dim = projection_dim
        if projection_dim is not None:
            self.r_trans = nn.Linear(hidden_size // 2 * d, projection_dim // 2)
            self.i_trans = nn.Linear(hidden_size // 2 * d, projection_dim // 2)

    def forward(self, inputs):
        real, imag = inputs
        r2r = self.real_lstm(real)[0]
        r2i = self.imag_lstm(real)[0]
        i2r = self.real_lstm(imag)[0]
        i2i = self.imag_lstm(imag)[0]
        real_out = r2r - i2i
        imag_out = i2r + r2i
        if self.projection_dim is not None:
            real_out = self.r_trans(real_out)
            imag_out = self.i_trans(imag_out)
        return [real_out, imag_out]

    def flatten_parameters(self):
        self.real_lstm.flatten_parameters()
        self.imag_lstm.flatten_parameters()


def complex_cat(inputs, axis):
    real, imag = [], []
    for data in inputs:
        r, i = torch.chunk(data, 2, axis)
        real.append(r)
        imag.append(i)
    return torch.cat([torch.cat(real, axis), torch.cat(imag, axis)], axis)


class ComplexBatchNorm(nn.Module):        # imported by DCCRN_cprs.py:6, unused with use_cbn=False
    def __init__(self, *a, **k):
        super().__init__()
        raise NotImplementedError("use_cbn=True is not on the decode path (dccrn_decode_vb.py:11)")
