"""Test-only helpers.  `OracleModel` has the product's parameter tree (so state_dicts interchange) but
computes its forward with the CPU oracle - it exists so that the loader / training loop / data-parallel
plumbing can be exercised on a machine without a GPU.  Never imported by the product."""
import torch

from drin_amd.config import DrinConfig
from drin_amd.model import Model
from oracle import drin_oracle as O


class OracleModel(Model):
    def forward(self, batch):
        p = dict(self.named_parameters())
        cfg: DrinConfig = self.cfg
        return O.forward(p, batch, num_layers=cfg.num_gcn_layers, edge_enabled=cfg.gcn_edge_enabled,
                         dynamic=cfg.gcn_edge_type == "dynamic", vector=cfg.gcn_edge_feature == "vector",
                         vertex_activation=cfg.gcn_vertex_activation, edge_activation=cfg.gcn_edge_activation)
