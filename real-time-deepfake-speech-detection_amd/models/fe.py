"""SSL frontend wrappers -- mirror of the reference's models/fe.py:8-40 (XLSR_FE),
:43-50 (middle_indices) and :53-99 (My_XLSR_FE)."""
import torch
from torch import nn

from afx.host import AfxModule, Wav2Vec2Trunk, resolve_device, ssl_checkpoint_or_synthetic

__all__ = ["XLSR_FE", "My_XLSR_FE", "middle_indices", "nn", "torch"]


class XLSR_FE(AfxModule):
    """models/fe.py:8-40.  The reference loads fairseq's xlsr2_300m.pt from a hard-coded
    path and ignores ``ssl_cpkt_path``; here the path is honoured when the
    file exists, otherwise the trunk keeps its seeded synthetic weights."""

    afx_arch = "ssl"

    def __init__(self, device, ssl_cpkt_path=None, num_layers=24):
        super().__init__()
        self.model = Wav2Vec2Trunk(num_layers)
        ssl_checkpoint_or_synthetic(self.model, ssl_cpkt_path)
        self.model = self.model.to(resolve_device(device))
        self.out_dim = 1024

    def _afx_trunk(self):
        return self.model

    def extract_feat(self, input_data):
        # models/fe.py:17-21: (B,L,1) inputs use channel 0; returns (B,T,1024)
        input_tmp = input_data[:, :, 0] if input_data.ndim == 3 else input_data
        self._afx_check(input_tmp)
        return self._afx_engine().ssl(input_tmp)

    def forward(self, input_data):
        return self.extract_feat(input_data)

    def partial_freeze_layers(self, target_layers: list, non_target_layers: list):
        # models/fe.py:26-35 (training-time helper; kept for constructor compatibility)
        for name, param in self.model.named_parameters():
            if any(layer in name for layer in target_layers) and not any(layer in name for layer in non_target_layers):
                param.requires_grad = False
        self.random_init_layers(non_target_layers)

    def random_init_layers(self, target_layers: list):
        # models/fe.py:36-40
        for name, param in self.model.named_parameters():
            if any(layer in name for layer in target_layers) and param.dim() >= 2:
                torch.nn.init.xavier_uniform_(param)


def middle_indices(array_length, number_of_middle_elements):
    """models/fe.py:43-50."""
    start_index = (array_length - number_of_middle_elements) // 2
    return list(range(start_index, start_index + number_of_middle_elements))


class My_XLSR_FE(XLSR_FE):
    """models/fe.py:53-99: the truncated (distilled-student) trunk."""

    def __init__(self, device, **kwargs):
        num_layers = kwargs.get("num_layers", 24)
        order = kwargs.get("order", "first")
        custom_order = kwargs.get("custom_order", None)
        if num_layers < 1 or num_layers > 24:
            raise ValueError("Number of layers must be at least 1 and at most 24.")
        if order not in ("last", "first", "middle"):
            if custom_order is None:
                raise ValueError("Custom order must be provided as a list of integers (0-23).")
            if type(custom_order) != list:
                raise ValueError("Custom order must be a list of integers.")
        super().__init__(device, ssl_cpkt_path=kwargs.get("ssl_cpkt_path"))
        self.num_layers, self.order, self.custom_order = num_layers, order, custom_order
        layers = self.model.encoder.layers
        if order == "last":
            self.model.encoder.layers = layers[-num_layers:]
        elif order == "first":
            self.model.encoder.layers = layers[:num_layers]
        elif order == "middle":
            self.model.encoder.layers = nn.ModuleList([layers[i] for i in middle_indices(24, num_layers)])
        else:
            self.model.encoder.layers = nn.ModuleList([layers[i] for i in custom_order])
