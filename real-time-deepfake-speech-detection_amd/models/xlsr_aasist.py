"""XLS-R + AASIST -- mirror of the reference's models/xlsr_aasist.py: XLSR_AASIST
(:5-177) and My_XLSR_AASIST (:180-339).  Same constructor signature, sub-module names
and state_dict keys; ``forward(x) -> (B,2)`` logits run as one native call
(afx_forward): SSL trunk on the fp16/bf16 matrix cores, back-end in fp32."""
import torch
import torch.nn as nn

from afx.host import AfxModule

from .aasist_modules import *  # noqa: F401,F403
from .aasist_modules import GraphAttentionLayer, GraphPool, HtrgGraphAttentionLayer, Residual_block
from .fe import *  # noqa: F401,F403
from .fe import My_XLSR_FE, XLSR_FE


class _AasistBase(AfxModule):
    afx_arch = "xlsr_aasist"

    def _build_head(self, kwargs):
        partial_freeze_layers = kwargs.get("partial_freeze_layers", None)
        partial_freeze_init_layers = kwargs.get("partial_freeze_init_layers", [])
        if partial_freeze_layers:  # models/xlsr_aasist.py:13-18
            self.ssl_model.partial_freeze_layers(partial_freeze_layers.get("target_layers", []),
                                                 partial_freeze_layers.get("non_target_layers", []))
        if len(partial_freeze_init_layers) > 0:
            self.ssl_model.random_init_layers(partial_freeze_init_layers)
        # AASIST parameters (models/xlsr_aasist.py:23-84)
        filts = [128, [1, 32], [32, 32], [32, 64], [64, 64]]
        gat_dims = [64, 32]
        pool_ratios = [0.5, 0.5, 0.5, 0.5]
        temperatures = [2.0, 2.0, 100.0, 100.0]
        self.LL = nn.Linear(self.ssl_model.out_dim, 128)
        self.first_bn = nn.BatchNorm2d(num_features=1)
        self.first_bn1 = nn.BatchNorm2d(num_features=64)
        self.drop = nn.Dropout(0.5, inplace=True)
        self.drop_way = nn.Dropout(0.2, inplace=True)
        self.selu = nn.SELU(inplace=True)
        self.encoder = nn.Sequential(
            nn.Sequential(Residual_block(nb_filts=filts[1], first=True)),
            nn.Sequential(Residual_block(nb_filts=filts[2])),
            nn.Sequential(Residual_block(nb_filts=filts[3])),
            nn.Sequential(Residual_block(nb_filts=filts[4])),
            nn.Sequential(Residual_block(nb_filts=filts[4])),
            nn.Sequential(Residual_block(nb_filts=filts[4])))
        self.attention = nn.Sequential(nn.Conv2d(64, 128, kernel_size=(1, 1)), nn.SELU(inplace=True),
                                       nn.BatchNorm2d(128), nn.Conv2d(128, 64, kernel_size=(1, 1)))
        self.pos_S = nn.Parameter(torch.randn(1, 42, filts[-1][-1]))
        self.master1 = nn.Parameter(torch.randn(1, 1, gat_dims[0]))
        self.master2 = nn.Parameter(torch.randn(1, 1, gat_dims[0]))
        self.GAT_layer_S = GraphAttentionLayer(filts[-1][-1], gat_dims[0], temperature=temperatures[0])
        self.GAT_layer_T = GraphAttentionLayer(filts[-1][-1], gat_dims[0], temperature=temperatures[1])
        self.HtrgGAT_layer_ST11 = HtrgGraphAttentionLayer(gat_dims[0], gat_dims[1], temperature=temperatures[2])
        self.HtrgGAT_layer_ST12 = HtrgGraphAttentionLayer(gat_dims[1], gat_dims[1], temperature=temperatures[2])
        self.HtrgGAT_layer_ST21 = HtrgGraphAttentionLayer(gat_dims[0], gat_dims[1], temperature=temperatures[2])
        self.HtrgGAT_layer_ST22 = HtrgGraphAttentionLayer(gat_dims[1], gat_dims[1], temperature=temperatures[2])
        self.pool_S = GraphPool(pool_ratios[0], gat_dims[0], 0.3)
        self.pool_T = GraphPool(pool_ratios[1], gat_dims[0], 0.3)
        self.pool_hS1 = GraphPool(pool_ratios[2], gat_dims[1], 0.3)
        self.pool_hT1 = GraphPool(pool_ratios[2], gat_dims[1], 0.3)
        self.pool_hS2 = GraphPool(pool_ratios[2], gat_dims[1], 0.3)
        self.pool_hT2 = GraphPool(pool_ratios[2], gat_dims[1], 0.3)
        self.out_layer = nn.Linear(5 * gat_dims[1], 2)

    def _afx_trunk(self):
        return self.ssl_model.model

    def forward(self, x):
        # models/xlsr_aasist.py:86-177 in one native call; x is (B,L) or (B,L,1)
        x = x.squeeze(-1) if x.ndim == 3 else x
        self._afx_check(x)
        return self._afx_engine().forward(x)


class XLSR_AASIST(_AasistBase):
    def __init__(self, device, ssl_cpkt_path=None, **kwargs) -> None:
        super().__init__()
        self.ssl_model = XLSR_FE(device, ssl_cpkt_path=ssl_cpkt_path)
        self._build_head(kwargs)


class My_XLSR_AASIST(_AasistBase):
    def __init__(self, device, ssl_cpkt_path=None, **kwargs) -> None:
        super().__init__()
        self.ssl_model = My_XLSR_FE(device, ssl_cpkt_path=ssl_cpkt_path, **kwargs)
        self._build_head(kwargs)
