"""Golden-case definitions shared by `oracle/gen_golden.py` (which runs the reference) and
the tests (which only read the committed fixtures).  TEST INFRASTRUCTURE."""
from __future__ import annotations

from drin_amd import synth
from drin_amd.config import DrinConfig

TINY = dict(bert_embed_dim=64, gcn_embed_dim=64, resnet_embed_dim=128, max_mention_sentence_len=12, resnet_num_region=5)

CASES = {
    # name: (cfg, batch, data_seed, weight_seed, full_entity_vertices, with_grads)
    "wd_b4": (DrinConfig(), 4, 1, 7, False, True),
    "wm_b2": (DrinConfig(dataset_name="wikimel", num_candidates_data=100, max_entity_attr_token_len=8,
                         max_mention_sentence_len=16, resnet_num_region=4), 2, 2, 7, False, True),
    "tiny_wd": (DrinConfig(**TINY), 5, 3, 8, True, True),
    "tiny_wm": (DrinConfig(dataset_name="wikimel", num_candidates_data=6, max_entity_attr_token_len=6, **TINY), 3, 4, 8, True, True),
    "tiny_wd_edges_1010": (DrinConfig(gcn_edge_enabled=(1, 0, 1, 0), **TINY), 3, 5, 8, True, True),
    "tiny_wd_static": (DrinConfig(gcn_edge_type="static", **TINY), 3, 6, 8, True, True),
    "tiny_wd_layers1": (DrinConfig(num_gcn_layers=1, **TINY), 3, 7, 8, True, True),
    "tiny_wd_layers3": (DrinConfig(num_gcn_layers=3, **TINY), 3, 8, 8, True, True),
    "tiny_wd_vector": (DrinConfig(gcn_edge_feature="vector", **TINY), 3, 12, 8, True, True),
    "tiny_wm_vector_static": (DrinConfig(dataset_name="wikimel", num_candidates_data=6, max_entity_attr_token_len=6,
                                         gcn_edge_feature="vector", gcn_edge_type="static", **TINY), 2, 13, 8, True, True),
    # activations by name (args.py:35-36; model.py:117-118 takes getattr(torch.nn.functional, name))
    "tiny_wd_relu_tanh": (DrinConfig(gcn_vertex_activation="relu", gcn_edge_activation="tanh", **TINY), 3, 14, 8, True, True),
    "tiny_wm_silu_relu": (DrinConfig(dataset_name="wikimel", num_candidates_data=6, max_entity_attr_token_len=6,
                                     gcn_vertex_activation="silu", gcn_edge_activation="relu", **TINY), 3, 15, 8, True, True),
    "tiny_wd_tanh_vector": (DrinConfig(gcn_vertex_activation="tanh", gcn_edge_activation="tanh", gcn_edge_feature="vector", **TINY),
                            3, 16, 8, True, True),
    # (forward only: with all vertex features in (0, 1) the score gradients nearly cancel in the LayerNorm column sums - the
    #  reference's own fp32 autograd is only good to ~1e-3 there; the backward of this activation is covered against the
    #  oracle by the random-geometry sweep)
    "tiny_wd_sigmoid_vertex": (DrinConfig(gcn_vertex_activation="sigmoid", num_gcn_layers=3, **TINY), 2, 17, 8, True, False),
    # edge activations without a derivative-from-output (the backward keeps the pre-activation for them)
    "tiny_wd_gelu_edge": (DrinConfig(gcn_edge_activation="gelu", num_gcn_layers=3, **TINY), 3, 18, 8, True, True),
    "tiny_wm_silu_edge_vector": (DrinConfig(dataset_name="wikimel", num_candidates_data=6, max_entity_attr_token_len=6,
                                            gcn_edge_activation="silu", gcn_edge_feature="vector", **TINY), 3, 19, 8, True, True),
    "tiny_wm_n37": (DrinConfig(dataset_name="wikimel", num_candidates_data=36, max_entity_attr_token_len=9, **TINY), 4, 9, 8, True, False),
}


def edge_case_batches(name, cfg, batch):
    """Hand-edited inputs for the reference's own corner semantics."""
    if name == "tiny_wd":
        # span of length 1, span reaching the last token, an all-zero object score row
        batch[2][0], batch[3][0] = 3, 4
        batch[2][1], batch[3][1] = cfg.max_mention_sentence_len - 2, cfg.max_mention_sentence_len
        batch[6][2].zero_()
        batch[11][3].zero_()
    if name == "tiny_wm":
        # ntok = 3 -> exactly one pooled token; ntok = T
        batch[8][0, 0] = 0
        batch[8][0, 0, :3] = 1
        batch[8][1, 2] = 1
    return batch




def build_case(name):
    """(cfg, state_dict, 15-sequence batch) of a golden case, regenerated from its seeds."""
    cfg, B, dseed, wseed, _full, _grads = CASES[name]
    sd = synth.make_state_dict(cfg, wseed)
    batch = edge_case_batches(name, cfg, synth.make_batch(cfg, B, dseed))
    return cfg, sd, batch
