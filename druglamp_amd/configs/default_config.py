"""Configuration surface of the reference (configs/default_config.py:4-89, configs/*.yaml) without
yacs: the same key tree (DRUG / PROTEIN / DECODER / SOLVER / RESULT / RS / COMET), the same defaults,
`get_cfg_defaults()`, `get_model_defaults(hidden)` and yaml merging (`merge_from_file`)."""
from __future__ import annotations

import os

import yaml


class CfgNode(dict):
    """Attribute-style nested dict with the handful of yacs methods the reference uses."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __setattr__(self, k, v):
        self[k] = v

    def clone(self):
        out = CfgNode()
        for k, v in self.items():
            out[k] = v.clone() if isinstance(v, CfgNode) else (list(v) if isinstance(v, list) else v)
        return out

    def merge_from_dict(self, d):
        for k, v in d.items():
            if k not in self:
                raise KeyError("Non-existent config key: %s" % k)
            if isinstance(v, dict):
                self[k].merge_from_dict(v)
            else:
                old = self[k]
                if isinstance(old, float) and isinstance(v, (int, str)):
                    v = float(v)           # yaml reads 1e-4 as a string without a dot
                self[k] = v

    def merge_from_file(self, path):
        with open(path) as f:
            self.merge_from_dict(yaml.safe_load(f) or {})

    def freeze(self):
        return None


CN = CfgNode


def get_cfg_defaults() -> CfgNode:
    c = CN()
    c.DRUG = CN(NODE_IN_FEATS=75, MAX_NODES=512, PADDING=True)
    c.PROTEIN = CN(KERNEL_SIZE=[3, 6, 9], PADDING=True, SEQ_LEN=9 * 256, SITE_LEN=9)
    c.DECODER = CN(NAME="MLP", IN_DIM=256, HIDDEN_DIM=512, OUT_DIM=128, BINARY=-1)
    c.SOLVER = CN(MAX_EPOCH=-1, BATCH_SIZE=-1, NUM_WORKERS=-1, LR=-1., SSL_LR=-1., CM_LR=-1., SEED=-1)
    c.RESULT = CN(OUTPUT_DIR=f"{os.getcwd()}/results/")
    c.RS = CN(TASK=False, METHOD="2C2P", SSL=False, CM=False, INIT_EPOCH=-1, EPOCH_STEP=-1, MAX_MARGIN=-1.,
              RESET_EPOCH=-1,
              GLOBAL_BATCH=False,   # NEW (not in the reference): the batch-level heads see the all-gathered GLOBAL batch at
                                    # world > 1 — cross-modal triplets (CM) and the NT-Xent denominator (SSL, simclr)
              DRUG_SSL_TYPE="simsiam")   # NEW: the reference hard-codes 'simsiam' (basic_model.py:85), which leaves its
                                         # nt_xent_loss unreachable; "simclr" selects it (SSL.drug_simclr, :35-41)
    c.COMET = CN(WORKSPACE="lzcstan", PROJECT_NAME="DrugLAMP", USE=True, TAG="Reproduce")
    return c


def get_lamp_config(hidden_size: int, feat_len: int = 256) -> CfgNode:
    """PMMA hyper-parameters (default_config.py:67-84).  `feat_len` (the number of protein sites, = the length of
    both PMMA streams and of the positional tables, embed.py:32-33) is hard-coded to 256 in the reference; here it
    follows PROTEIN.SEQ_LEN // PROTEIN.SITE_LEN (default 2304 // 9 = 256) so that long-protein configurations
    (BASELINE config 5: SEQ_LEN = 9216 -> 1024 sites) are a yaml change."""
    c = CN()
    c.n_output = 1
    c.hidden_size = hidden_size * 2
    c.num_features_llm = c.hidden_size
    c.mlha_dropout = 0
    c.transformer = CN(num_heads=4, num_p_plus_s_layers=4, attention_dropout_rate=0, dropout_rate=0.1)
    c.classifier = "token"
    c.representation_size = None
    c.mol_len = 512
    c.feat_len = int(feat_len)
    return c


def get_model_defaults(hidden_size: int, feat_len: int = 256) -> CfgNode:
    c = get_lamp_config(hidden_size, feat_len)
    c.mol_len = c.feat_len
    return c


def load_yaml_into(cfg: CfgNode, model_name: str) -> CfgNode:
    """cfg.merge_from_file(configs/{model}.yaml) as main.py:57-59 does."""
    cfg.merge_from_file(os.path.join(os.path.dirname(os.path.abspath(__file__)), model_name + ".yaml"))
    return cfg
