"""Mirror of the reference's models/models.py: SSLModel (:13-41) plus the same four
AASIST modules it carries as a copy (:49-432; here re-exported from aasist_modules)."""
import logging

from afx.host import AfxModule, Wav2Vec2Trunk, resolve_device, ssl_checkpoint_or_synthetic

from .aasist_modules import GraphAttentionLayer, GraphPool, HtrgGraphAttentionLayer, Residual_block  # noqa: F401


class SSLModel(AfxModule):
    """models/models.py:13-41: like XLSR_FE but with the checkpoint path as an argument."""

    afx_arch = "ssl"

    def __init__(self, device, cp_path, out_dim):
        super().__init__()
        self.model = Wav2Vec2Trunk(24)
        ssl_checkpoint_or_synthetic(self.model, cp_path)
        self.model = self.model.to(resolve_device(device))
        self.out_dim = out_dim
        self.freeze = False

    def _afx_trunk(self):
        return self.model

    def extract_feat(self, input_data):
        input_tmp = input_data[:, :, 0] if input_data.ndim == 3 else input_data
        self._afx_check(input_tmp)
        return self._afx_engine().ssl(input_tmp)

    def forward(self, input_data):
        return self.extract_feat(input_data)

    def frozen(self):
        logging.info("Freezing the model")
        for param in self.model.parameters():
            param.requires_grad = False
        self.freeze = True

    def unfrozen(self):
        logging.info("Unfreezing the model")
        for param in self.model.parameters():
            param.requires_grad = True
        self.freeze = False
