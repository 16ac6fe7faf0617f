"""AASIST graph modules -- mirror of the reference's models/aasist_modules.py:
GraphAttentionLayer (:17-110), HtrgGraphAttentionLayer (:112-294), GraphPool
(:296-338), Residual_block (:340-397).  Same constructors and state_dict keys; the
eval forward of the three graph modules runs on the fp32 HIP kernels through the C
ABI (afx_k_gat / afx_k_hgat / afx_k_graph_pool).  Inside XLSR_AASIST the whole
back-end runs as one fused native call, so these forwards only serve standalone use.
"""
from typing import Optional, Union

import torch
import torch.nn.functional as F  # noqa: F401  (re-exported: the reference's star-import provides F)
from torch import nn

from afx import kernels as _K

__all__ = ["GraphAttentionLayer", "HtrgGraphAttentionLayer", "GraphPool", "Residual_block", "nn", "torch", "F",
           "Optional", "Union"]


def _eval_only(mod, x):
    if mod.training:
        raise RuntimeError("the MI355X-native graph kernels are inference-only: call .eval() first")
    if not x.is_cuda:
        raise RuntimeError("input must be on the GPU: the native path has no CPU fallback")


def _new_params(*size):
    out = nn.Parameter(torch.empty(*size))
    nn.init.xavier_normal_(out)
    return out


def _bn(bn):
    return _K.bn_fold(bn.weight.detach(), bn.bias.detach(), bn.running_mean, bn.running_var, bn.eps)


class GraphAttentionLayer(nn.Module):
    def __init__(self, in_dim, out_dim, **kwargs):
        super().__init__()
        self.att_proj = nn.Linear(in_dim, out_dim)
        self.att_weight = _new_params(out_dim, 1)
        self.proj_with_att = nn.Linear(in_dim, out_dim)
        self.proj_without_att = nn.Linear(in_dim, out_dim)
        self.bn = nn.BatchNorm1d(out_dim)
        self.input_drop = nn.Dropout(p=0.2)
        self.act = nn.SELU(inplace=True)
        self.temp = kwargs.get("temperature", 1.0)

    def forward(self, x):
        _eval_only(self, x)
        sc, sh = _bn(self.bn)
        p = dict(att_w=self.att_proj.weight, att_b=self.att_proj.bias, att_vec=self.att_weight.reshape(-1),
                 w1=self.proj_with_att.weight, b1=self.proj_with_att.bias, w2=self.proj_without_att.weight,
                 b2=self.proj_without_att.bias, bn_scale=sc, bn_shift=sh)
        return _K.gat(x.float().contiguous(), {k: v.detach().contiguous() for k, v in p.items()}, float(self.temp))


class HtrgGraphAttentionLayer(nn.Module):
    def __init__(self, in_dim, out_dim, **kwargs):
        super().__init__()
        self.proj_type1 = nn.Linear(in_dim, in_dim)
        self.proj_type2 = nn.Linear(in_dim, in_dim)
        self.att_proj = nn.Linear(in_dim, out_dim)
        self.att_projM = nn.Linear(in_dim, out_dim)
        self.att_weight11 = _new_params(out_dim, 1)
        self.att_weight22 = _new_params(out_dim, 1)
        self.att_weight12 = _new_params(out_dim, 1)
        self.att_weightM = _new_params(out_dim, 1)
        self.proj_with_att = nn.Linear(in_dim, out_dim)
        self.proj_without_att = nn.Linear(in_dim, out_dim)
        self.proj_with_attM = nn.Linear(in_dim, out_dim)
        self.proj_without_attM = nn.Linear(in_dim, out_dim)
        self.bn = nn.BatchNorm1d(out_dim)
        self.input_drop = nn.Dropout(p=0.2)
        self.act = nn.SELU(inplace=True)
        self.temp = kwargs.get("temperature", 1.0)

    def forward(self, x1, x2, master: Optional[torch.Tensor] = None):
        _eval_only(self, x1)
        sc, sh = _bn(self.bn)
        p = dict(t1w=self.proj_type1.weight, t1b=self.proj_type1.bias, t2w=self.proj_type2.weight,
                 t2b=self.proj_type2.bias, att_w=self.att_proj.weight, att_b=self.att_proj.bias,
                 attM_w=self.att_projM.weight, attM_b=self.att_projM.bias, v11=self.att_weight11.reshape(-1),
                 v22=self.att_weight22.reshape(-1), v12=self.att_weight12.reshape(-1),
                 vM=self.att_weightM.reshape(-1), w1=self.proj_with_att.weight, b1=self.proj_with_att.bias,
                 w2=self.proj_without_att.weight, b2=self.proj_without_att.bias, w1M=self.proj_with_attM.weight,
                 b1M=self.proj_with_attM.bias, w2M=self.proj_without_attM.weight, b2M=self.proj_without_attM.bias,
                 bn_scale=sc, bn_shift=sh)
        m = None if master is None else master.detach().float().reshape(master.shape[0], -1)
        return _K.hgat(x1.float().contiguous(), x2.float().contiguous(),
                       {k: v.detach().contiguous() for k, v in p.items()}, float(self.temp), master=m)


class GraphPool(nn.Module):
    def __init__(self, k: float, in_dim: int, p: Union[float, int]):
        super().__init__()
        self.k = torch.tensor(k)
        self.sigmoid = nn.Sigmoid()
        self.proj = nn.Linear(in_dim, 1)
        self.drop = nn.Dropout(p=p) if p > 0 else nn.Identity()
        self.in_dim = in_dim

    def forward(self, h):
        _eval_only(self, h)
        return _K.graph_pool(h.float().contiguous(), self.proj.weight.detach().reshape(-1).contiguous(),
                             self.proj.bias.detach().contiguous(), float(self.k))


class Residual_block(nn.Module):
    """models/aasist_modules.py:340-397.  bn1 is present in checkpoints but never reaches the
    output (Q2: the reference overwrites the bn1 + SELU result with conv1(x)).  Inside
    XLSR_AASIST the six blocks run fused in the native back-end; stand-alone, ``forward``
    runs the same kernels on one NCHW image (afx_k_resblock)."""

    def __init__(self, nb_filts, first=False):
        super().__init__()
        self.first = first
        self.bn1 = None
        self.conv_downsample = None
        if not self.first:
            self.bn1 = nn.BatchNorm2d(num_features=nb_filts[0])
        self.conv1 = nn.Conv2d(nb_filts[0], nb_filts[1], kernel_size=(2, 3), padding=(1, 1), stride=1)
        self.selu = nn.SELU(inplace=True)
        self.bn2 = nn.BatchNorm2d(num_features=nb_filts[1])
        self.conv2 = nn.Conv2d(nb_filts[1], nb_filts[1], kernel_size=(2, 3), padding=(0, 1), stride=1)
        if nb_filts[0] != nb_filts[1]:
            self.downsample = True
            self.conv_downsample = nn.Conv2d(nb_filts[0], nb_filts[1], padding=(0, 1), kernel_size=(1, 3), stride=1)
        else:
            self.downsample = False

    def forward(self, x):
        _eval_only(self, x)
        d = lambda t: t.detach().float().contiguous()
        sc, sh = _bn(self.bn2)
        dw = d(self.conv_downsample.weight) if self.downsample else None
        db = d(self.conv_downsample.bias) if self.downsample else None
        return _K.resblock(x.float().contiguous(), d(self.conv1.weight), d(self.conv1.bias), sc, sh,
                           d(self.conv2.weight), d(self.conv2.bias), dw, db)
